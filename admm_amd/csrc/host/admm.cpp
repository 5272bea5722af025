// Driver with the reference's loop (admm.cpp:41-125): set the partitioner and
// the penalties, describe the problem, initialise, then per iteration Compute,
// Communicate, Update and the objective, and print the same timing labels.
// Input is a graph file named by argv[1], as in the reference (admm.cpp:30),
// or a synthetic grid.
//
//   admm <graph file> [--solver edge|chain] [--iters N] [--strips G] [--quiet] [--async]
//   admm <rows> <cols> <k> [--solver edge|chain] [--iters N] [--pairwise potts|trunc_linear|dense]
//        [--seed S] [--strips G] [--batch B] [--quiet] [--save FILE] [--tree N] [--async]
//
// The three phases are queued on the device and return at once; so that
// maxCompTime, maxCommTime and maxUpdateTime mean what the reference's mean (its
// PROFILE macro puts a barrier before each reading, admm.cpp:15-18, 79-91) every
// strip is waited for before a timer is read.  --async leaves the waits out: the
// labels then time the enqueueing only and the device time shows up in optTime.
//
// --tree N prints, after the last iteration, the variables of tree N of strip 0
// under the names of the reference's class Edge (Tree.h:26-35).
//
// --save FILE writes the synthetic problem as a graph file (the reference's
// GENERATE build, admm.cpp:32-39) before solving it.
// --strips G cuts the grid into G row/column strips held by this one process,
// strip g on device g modulo the device count.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "Grid.h"

static double now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char *argv[]) {
  const bool fromFile = argc >= 2 && (argv[1][0] < '0' || argv[1][0] > '9');
  if (argc < 2 || (!fromFile && argc < 4)) {
    fprintf(stderr, "usage: %s <graph file> [--solver edge|chain] [--iters N] [--strips G] [--quiet]\n"
                    "       %s <rows> <cols> <k> [--solver edge|chain] [--iters N] [--pairwise KIND] [--seed S] [--strips G] [--batch B] [--quiet] [--save FILE]\n",
            argv[0], argv[0]);
    return 2;
  }
  const std::string fname = fromFile ? argv[1] : "";
  const int rows = fromFile ? 0 : atoi(argv[1]), cols = fromFile ? 0 : atoi(argv[2]), k = fromFile ? 0 : atoi(argv[3]);
  std::string solver = "edge", pairwise = "potts", saveTo;
  int totalRuns = 500, strips = 1, batch = 1, showTree = -1;  // admm.cpp:74
  unsigned seed = 1234;
  bool quiet = false, async = false;
  for (int i = fromFile ? 2 : 4; i < argc; i++) {
    auto val = [&](const char *n) -> const char * {
      if (i + 1 >= argc) { fprintf(stderr, "%s needs a value\n", n); exit(2); }
      return argv[++i];
    };
    if (!strcmp(argv[i], "--solver")) solver = val("--solver");
    else if (!strcmp(argv[i], "--iters")) totalRuns = atoi(val("--iters"));
    else if (!strcmp(argv[i], "--pairwise")) pairwise = val("--pairwise");
    else if (!strcmp(argv[i], "--seed")) seed = (unsigned)strtoul(val("--seed"), nullptr, 10);
    else if (!strcmp(argv[i], "--strips")) strips = atoi(val("--strips"));
    else if (!strcmp(argv[i], "--batch")) batch = atoi(val("--batch"));
    else if (!strcmp(argv[i], "--save")) saveTo = val("--save");
    else if (!strcmp(argv[i], "--tree")) showTree = atoi(val("--tree"));
    else if (!strcmp(argv[i], "--quiet")) quiet = true;
    else if (!strcmp(argv[i], "--async")) async = true;
    else { fprintf(stderr, "unknown option %s\n", argv[i]); return 2; }
  }
  try {
    const double startRun = now();
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { fprintf(stderr, "no HIP device\n"); return 1; }
    std::vector<std::unique_ptr<Grid>> grids;
    std::vector<Grid *> ptrs;
    double t = now();
    for (int g = 0; g < strips; g++) {
      grids.emplace_back(new Grid(g, strips, g % ndev));
      Grid &grid = *grids.back();
      grid.SetPartitioner(solver == "edge" ? EDGE : ROW);  // admm.cpp:41
      grid.SetPenalties(0.5, 1.0);                          // admm.cpp:44
      if (fromFile) grid.SetGlobalInfo(fname);               // admm.cpp:66
      else grid.SetGlobalInfo(rows, cols, k, k * k);
      grid.SetBatch(batch);
      ptrs.push_back(&grid);
    }
    const double allocTime = now() - t; t = now();
    for (auto &g : grids) {
      if (fromFile) g->Load(fname);                          // admm.cpp:70
      else g->Generate(seed, pairwise);
    }
    if (!saveTo.empty()) grids[0]->Save(saveTo);
    const double loadTime = now() - t; t = now();
    for (auto &g : grids) g->InitOptimization();
    const double initTime = now() - t; t = now();
    double compTime = 0, commTime = 0, updateTime = 0;
    auto settle = [&]() { if (!async) for (auto &g : grids) g->Synchronize(); };
    int irun = 0;
    while (irun++ < totalRuns) {
      double it = now();
      for (auto &g : grids) g->Compute();
      settle();
      compTime += now() - it; it = now();
      if (strips > 1) Grid::CommunicateGroup(ptrs);
      else grids[0]->Communicate();
      settle();
      commTime += now() - it; it = now();
      for (auto &g : grids) g->Update();
      settle();
      updateTime += now() - it;
      double totalObj = 0.0;
      for (auto &g : grids) totalObj += g->ComputeLinearObj();  // admm.cpp:94-95
      if (!quiet) fprintf(stderr, "totalObj:       %lf\n", totalObj);
    }
    if (solver != "edge") {
      double primal = 0, dual = 0;
      for (auto &g : grids) { double p, d; g->Objective(&p, &d); primal += p; dual += d; }
      fprintf(stderr, "primal:          %lf\ndual:            %lf\n", primal, dual);
    }
    if (showTree >= 0) {
      TreeSnapshot snap;
      grids[0]->Trees(snap);
      auto row = [](const char *name, const float *v, int n) {
        fprintf(stderr, "%s:", name);
        for (int i = 0; i < n; i++) fprintf(stderr, " %.9g", v[i]);
        fprintf(stderr, "\n");
      };
      if ((size_t)showTree < snap.edgeVal().size()) {
        const Edge &e = snap.edgeVal()[(size_t)showTree];
        row("x_mu_node", e.x_mu_node, NUM_NODE * e.k);
        row("x_potential_node", e.x_potential_node, NUM_NODE * e.k);
        row("edge_mu_node", e.edge_mu_node, NUM_NODE * e.k);
        row("lambda_node", e.lambda_node, NUM_NODE * e.k);
        row("x_mu_edge", e.x_mu_edge, e.ksquare);
        row("x_potential_edge", e.x_potential_edge, e.ksquare);
      } else if ((size_t)showTree < snap.chains().size()) {
        const Chain &c = snap.chains()[(size_t)showTree];
        fprintf(stderr, "%s chain %d, x:", c.column ? "column" : "row", c.index);
        for (int t = 0; t < c.length; t++) fprintf(stderr, " %d", c.x[t * c.stride]);
        fprintf(stderr, "\n");
        row("lambda_node[0]", c.lambda_node, c.k);
      } else {
        fprintf(stderr, "no tree %d\n", showTree);
      }
    }
    for (auto &g : grids) g->FinalizeOptimization();
    const double optTime = now() - t;
    const double runTime = now() - startRun;
    // one process: max and min over ranks coincide (admm.cpp:105-110)
    fprintf(stderr, "processes:       %d\n", strips);
    fprintf(stderr, "allocTime:       %lf\n", allocTime);
    fprintf(stderr, "loadTime:        %lf\n", loadTime);
    fprintf(stderr, "initTime:        %lf\n", initTime);
    fprintf(stderr, "optTime:         %lf\n", optTime);
    fprintf(stderr, "maxCompTime:     %lf\n", compTime);
    fprintf(stderr, "maxCommTime:     %lf\n", commTime);
    fprintf(stderr, "maxUpdateTime:   %lf\n", updateTime);
    fprintf(stderr, "minCompTime:     %lf\n", compTime);
    fprintf(stderr, "minCommTime:     %lf\n", commTime);
    fprintf(stderr, "minUpdateTime:   %lf\n", updateTime);
    fprintf(stderr, "maxTotal:        %lf\n", runTime);
  } catch (const AdmmError &e) {
    fprintf(stderr, "admm: error %d: %s\n", e.code, e.what());
    return 1;
  }
  return 0;
}
