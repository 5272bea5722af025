// Reader and writer for the NetCDF classic file family (CDF-1, CDF-2 and
// CDF-5, the "64-bit data" format of PnetCDF) without PnetCDF or MPI: just
// the part of the format the reference's graph files use (Grid.cpp:1066-1097):
// fixed-size dimensions, global attributes, non-record variables of type
// NC_INT / NC_FLOAT.  Everything in the file is big-endian.
//
// Layout (NetCDF classic format specification):
//   magic 'C' 'D' 'F' v   numrecs   dim_list   gatt_list   var_list   data
//   v = 1: every count and offset is 4 bytes;  v = 2: offsets ("begin") are
//   8 bytes;  v = 5: counts, dimension ids, sizes and offsets are 8 bytes.
//   list   = ABSENT (tag 0, count 0) | tag count item...
//   tags   : 0x0A dimensions, 0x0B variables, 0x0C attributes
//   name   = count bytes, zero-padded to a multiple of 4
//   dim    = name length
//   attr   = name nc_type count values (padded to a multiple of 4)
//   var    = name ndims dimid... attr_list nc_type vsize begin
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace ncfile {

enum NcType { NC_BYTE = 1, NC_CHAR = 2, NC_SHORT = 3, NC_INT = 4, NC_FLOAT = 5, NC_DOUBLE = 6,
              NC_UBYTE = 7, NC_USHORT = 8, NC_UINT = 9, NC_INT64 = 10, NC_UINT64 = 11 };

int type_size(int nc_type);  // 0 for an unknown type

struct Dim {
  std::string name;
  int64_t len = 0;  // 0 = the record dimension
};

struct Att {
  std::string name;
  int type = 0;
  int64_t nelems = 0;
  std::vector<unsigned char> raw;  // values as stored (big-endian), unpadded
  bool as_int(int *out) const;     // first value of an integer attribute
  std::string as_text() const;     // NC_CHAR attribute, trailing NULs dropped
};

struct Var {
  std::string name;
  std::vector<int64_t> dimids;
  std::vector<Att> atts;
  int type = 0;
  int64_t vsize = 0;
  int64_t begin = 0;
};

struct Header {
  int version = 0;  // 1, 2 or 5
  int64_t numrecs = 0;
  std::vector<Dim> dims;
  std::vector<Att> gatts;
  std::vector<Var> vars;
  const Att *find_att(const std::string &name) const;
  const Var *find_var(const std::string &name) const;
  int find_dim(const std::string &name) const;  // index or -1
};

// Every function returns true on success; on failure err holds the reason.

// Parses the header of an existing file.
bool read_header(const std::string &path, Header *h, std::string *err);

// Reads rows [start0, start0 + count0) of a two-dimensional non-record
// variable of 4-byte elements (NC_INT or NC_FLOAT) into host byte order.
// out holds count0 * dim1 elements.
bool read_rows(const std::string &path, const Header &h, const std::string &var, int64_t start0,
               int64_t count0, void *out, std::string *err);

// Writer: define, then begin_data (writes the header and fixes the layout),
// then write_rows in any order, then close.
class Writer {
 public:
  // version 1, 2 or 5; 0 picks 1 when every offset and size fits its 32-bit
  // fields (what ncmpi_create with no format flag writes, Grid.cpp:1067) and
  // 5 otherwise.
  explicit Writer(int version = 0) : want_version_(version) {}
  ~Writer();
  int def_dim(const std::string &name, int64_t len);
  int def_var(const std::string &name, int type, const std::vector<int> &dimids);
  void put_att_text(const std::string &name, const std::string &text, bool with_nul);
  void put_att_int(const std::string &name, int value);
  bool begin_data(const std::string &path, std::string *err);
  bool write_rows(int varid, int64_t start0, int64_t count0, const void *data, std::string *err);
  bool close(std::string *err);
  int version() const { return h_.version; }

 private:
  int want_version_;
  Header h_;
  int fd_ = -1;
};

}  // namespace ncfile
