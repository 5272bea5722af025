// Rectangle of grid rows / columns owned by one strip (the reference's BBox,
// BBox.h:12-34: upper-left corner and size, same accessor names).
#pragma once
#include <ostream>
#include <utility>

class BBox {
  std::pair<int, int> upperLeft{0, 0};
  std::pair<int, int> boxSize{0, 0};

 public:
  BBox() = default;
  BBox(int rowStart, int colStart, int rowSize, int colSize) { setBBox(rowStart, colStart, rowSize, colSize); }
  std::pair<int, int> &getUpperLeft() { return upperLeft; }
  std::pair<int, int> &getBoxSize() { return boxSize; }
  int getUpperLeftRow() const { return upperLeft.first; }
  int getUpperLeftCol() const { return upperLeft.second; }
  int getBoxRowSize() const { return boxSize.first; }
  int getBoxColSize() const { return boxSize.second; }
  void setUpperLeft(int rowStart, int colStart) { upperLeft = {rowStart, colStart}; }
  void setBoxSize(int rowSize, int colSize) { boxSize = {rowSize, colSize}; }
  void setBBox(int rowStart, int colStart, int rowSize, int colSize) {
    setUpperLeft(rowStart, colStart);
    setBoxSize(rowSize, colSize);
  }
  void setBBox(const int box[4]) { setBBox(box[0], box[1], box[2], box[3]); }
  friend std::ostream &operator<<(std::ostream &o, const BBox &b) {
    return o << "[(" << b.upperLeft.first << ", " << b.upperLeft.second << ") + (" << b.boxSize.first << ", "
             << b.boxSize.second << ")]";
  }
};
