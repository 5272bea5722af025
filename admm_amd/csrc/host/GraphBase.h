// Host-side graph container with the reference's class name and member names
// (GraphBase.h:11-24): potential counts and the two host potential arrays.
#pragma once
#include <string>
#include <vector>

class GraphBase {
 protected:
  int numVertPotentials = 0;  // labels per vertex (k)
  int numEdgePotentials = 0;  // entries per edge table (k*k)
  std::vector<float> vertPotentials;
  std::vector<float> edgePotentials;

 public:
  GraphBase() = default;
  GraphBase(int verticePotentialSize, int edgePotentialSize)
      : numVertPotentials(verticePotentialSize), numEdgePotentials(edgePotentialSize) {}
  virtual ~GraphBase() = default;
  virtual void Generate() {}
  virtual void Save(const std::string &) {}
  virtual void LoadByRow(const std::string &) {}
};
