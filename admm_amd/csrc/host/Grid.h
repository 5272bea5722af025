// C++ host with the reference's driver API (class Grid, Grid.h:98-137): the
// same method names, called in the same order as admm.cpp:41-101 calls them.
// Every method forwards to the C ABI of libadmm_hip.so (include/admm_hip.h);
// no arithmetic of the hot path happens here.
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/admm_graphfile.h"
#include "../../../include/admm_hip.h"
#include "BBox.h"
#include "GraphBase.h"
#include "Tree.h"

enum Partitioner { ROW, COL, BLOCK, EDGE };  // Grid.h:21

class AdmmError : public std::runtime_error {
 public:
  int code;
  AdmmError(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

class Grid : public GraphBase {
  int numGlobalRows = -1, numGlobalCols = -1;
  int numGlobalVerts = 0, numGlobalEdges = 0;
  int batch = 1;
  BBox bbox;              // this strip's rows
  float rhoEdge = 1.0f;   // Grid.cpp:79
  float penalty = 0.1f, penaltyPrime = 0.5f;  // Grid.cpp:80
  Partitioner part = EDGE;
  int rank = 0, numProcs = 1, device = 0;
  bool sharedTable = true;  // edgePotentials holds one k*k table, not one per edge
  std::vector<int> edgeList;  // 2 * numGlobalEdges vertex ids when the graph came from a file
  bool fromFile = false;
  // one strip of several, loaded from a grid file: the unaries stay in the file
  // until InitOptimization reads this strip's rows and columns, a bounded block
  // at a time, as a rank of the reference reads only its own hyperslab
  // (Grid.cpp:446-454, 510-523)
  std::string unaryFile;
  admm_ctx *ctx = nullptr;
  void UploadSharesFromFile();

  void check(int rc, const char *what) const;

 public:
  // rank / numProcs: which strip of how many this object is (the reference
  // reads them from MPI_Comm_rank / MPI_Comm_size, Grid.cpp:87-88).
  explicit Grid(int rank = 0, int numProcs = 1, int device = 0);
  ~Grid() override;
  Grid(const Grid &) = delete;
  Grid &operator=(const Grid &) = delete;

  void SetPenalties(float p, float pp) { penalty = p; penaltyPrime = pp; }
  void SetGlobalInfo(int globalRowSize, int globalColSize, int verticePotentialSize, int edgePotentialSize);
  // Sizes from a graph file's attributes (Grid.cpp:654-702).
  void SetGlobalInfo(const std::string &fname);
  void SetBatch(int n) { batch = n; }
  void SetPartitioner(Partitioner aPart);
  void SetRhoEdge(float aRhoEdge) { rhoEdge = aRhoEdge; }

  void Partition();
  // Synthetic potentials: reproducible LCG unaries in [0,1) and a named
  // pairwise table ("potts", "trunc_linear", "dense"); Grid::Generate of the
  // reference draws from rand() (Grid.cpp:631-645).
  void Generate(unsigned seed, const std::string &pairwise);
  void Generate() override { Generate(1234u, "potts"); }
  // Graph file -> edge list and potentials, sign-flipped into costs as the
  // reference's loader does (Grid.cpp:445-565).  EDGE: any graph.  ROW: the
  // file must hold a rows x cols grid in Generate's edge order with one
  // pairwise table common to all edges.
  void Load(const std::string &fname);
  // Potentials held by this object -> graph file, sign-flipped back into the
  // scores a file holds (Grid.cpp:1049-1131).
  void Save(const std::string &fname) override;
  // Raw binary files named by a manifest -> this object (Grid.cpp:985-1045);
  // follow with Save, as Converter.cpp does.
  void Convert(const std::string &manifest);
  // Potentials as the reference holds them after its loader's sign flip.
  void SetPotentials(const std::vector<float> &unary, const std::vector<float> &pairwise, bool perEdge);

  void InitOptimization();
  void Compute();
  void Communicate();
  void Update();
  double ComputeLinearObj();
  void FinalizeOptimization();
  // Waits for everything this strip has queued on its device.  The reference's phases
  // are synchronous; its PROFILE macro times them between barriers (admm.cpp:15-18).
  void Synchronize();

  // chain solver extras (no counterpart in the reference)
  void Objective(double *primal, double *dual);
  std::vector<int> Labels();
  // The trees' variables under the reference's names (class Edge, Tree.h:20-51):
  // a snapshot of the device state, waits for the stream.
  void Trees(TreeSnapshot &out);

  int GetRank() const { return rank; }
  int GetNprocs() const { return numProcs; }
  const BBox &GetBBox() const { return bbox; }
  admm_ctx *Handle() { return ctx; }
  // exchange between the strips of one problem held by this process
  static void CommunicateGroup(std::vector<Grid *> &strips);
};
