#include "NcFile.h"

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cstring>

namespace ncfile {

namespace {

constexpr uint32_t kTagDim = 0x0A, kTagVar = 0x0B, kTagAtt = 0x0C;
constexpr int64_t kDataAlign = 512;  // start of the data section; any multiple of 4 is valid

int64_t pad4(int64_t n) { return (n + 3) & ~int64_t(3); }

std::string sys_error(const std::string &what, const std::string &path) {
  return what + " '" + path + "': " + strerror(errno);
}

// ---- big-endian cursor over a file, read on demand ----
class In {
 public:
  In(int fd, int64_t size) : fd_(fd), size_(size) {}
  bool fail() const { return fail_; }
  int64_t pos() const { return pos_; }
  bool bytes(void *dst, int64_t n) {
    if (fail_ || n < 0 || pos_ + n > size_) { fail_ = true; return false; }
    char *p = static_cast<char *>(dst);
    int64_t done = 0;
    while (done < n) {
      const ssize_t r = pread(fd_, p + done, (size_t)(n - done), (off_t)(pos_ + done));
      if (r <= 0) { fail_ = true; return false; }
      done += r;
    }
    pos_ += n;
    return true;
  }
  uint32_t u32() {
    unsigned char b[4] = {0, 0, 0, 0};
    bytes(b, 4);
    return (uint32_t)b[0] << 24 | (uint32_t)b[1] << 16 | (uint32_t)b[2] << 8 | (uint32_t)b[3];
  }
  int64_t i64() {
    const uint64_t hi = u32(), lo = u32();
    return (int64_t)(hi << 32 | lo);
  }
  // a count / size / dimension id: 4 bytes in CDF-1 and CDF-2, 8 in CDF-5
  int64_t count(int version) { return version == 5 ? i64() : (int64_t)u32(); }
  bool name(int version, std::string *out) {
    const int64_t n = count(version);
    if (fail_ || n < 0 || n > (1 << 20)) { fail_ = true; return false; }
    std::string s((size_t)pad4(n), '\0');
    if (!bytes(&s[0], (int64_t)s.size())) return false;
    s.resize((size_t)n);
    *out = s;
    return true;
  }

 private:
  int fd_;
  int64_t size_, pos_ = 0;
  bool fail_ = false;
};

// ---- big-endian byte sink ----
class Out {
 public:
  std::vector<unsigned char> buf;
  void u32(uint32_t v) {
    buf.push_back((unsigned char)(v >> 24)); buf.push_back((unsigned char)(v >> 16));
    buf.push_back((unsigned char)(v >> 8)); buf.push_back((unsigned char)v);
  }
  void i64(int64_t v) { u32((uint32_t)((uint64_t)v >> 32)); u32((uint32_t)(uint64_t)v); }
  void count(int version, int64_t v) { if (version == 5) i64(v); else u32((uint32_t)v); }
  void raw(const void *p, size_t n) {
    const unsigned char *q = static_cast<const unsigned char *>(p);
    buf.insert(buf.end(), q, q + n);
    while (buf.size() & 3) buf.push_back(0);
  }
  void name(int version, const std::string &s) { count(version, (int64_t)s.size()); raw(s.data(), s.size()); }
};

bool read_atts(In &in, int version, std::vector<Att> *atts, std::string *err) {
  const uint32_t tag = in.u32();
  const int64_t n = in.count(version);
  if (in.fail() || (tag != kTagAtt && !(tag == 0 && n == 0))) { *err = "bad attribute list"; return false; }
  for (int64_t i = 0; i < n; i++) {
    Att a;
    if (!in.name(version, &a.name)) { *err = "truncated attribute name"; return false; }
    a.type = (int)in.u32();
    a.nelems = in.count(version);
    const int ts = type_size(a.type);
    if (in.fail() || ts == 0 || a.nelems < 0 || a.nelems > (1 << 26)) { *err = "bad attribute '" + a.name + "'"; return false; }
    std::vector<unsigned char> v((size_t)pad4(a.nelems * ts));
    if (!v.empty() && !in.bytes(v.data(), (int64_t)v.size())) { *err = "truncated attribute '" + a.name + "'"; return false; }
    v.resize((size_t)(a.nelems * ts));
    a.raw.swap(v);
    atts->push_back(a);
  }
  return true;
}

void write_atts(Out &o, int version, const std::vector<Att> &atts) {
  if (atts.empty()) { o.u32(0); o.count(version, 0); return; }
  o.u32(kTagAtt);
  o.count(version, (int64_t)atts.size());
  for (const Att &a : atts) {
    o.name(version, a.name);
    o.u32((uint32_t)a.type);
    o.count(version, a.nelems);
    o.raw(a.raw.data(), a.raw.size());
  }
}

void encode_header(const Header &h, Out &o) {
  const int v = h.version;
  o.buf.push_back('C'); o.buf.push_back('D'); o.buf.push_back('F'); o.buf.push_back((unsigned char)v);
  o.count(v, h.numrecs);
  if (h.dims.empty()) { o.u32(0); o.count(v, 0); }
  else {
    o.u32(kTagDim);
    o.count(v, (int64_t)h.dims.size());
    for (const Dim &d : h.dims) { o.name(v, d.name); o.count(v, d.len); }
  }
  write_atts(o, v, h.gatts);
  if (h.vars.empty()) { o.u32(0); o.count(v, 0); }
  else {
    o.u32(kTagVar);
    o.count(v, (int64_t)h.vars.size());
    for (const Var &x : h.vars) {
      o.name(v, x.name);
      o.count(v, (int64_t)x.dimids.size());
      for (int64_t id : x.dimids) o.count(v, id);
      write_atts(o, v, x.atts);
      o.u32((uint32_t)x.type);
      // CDF-1/2 cannot hold a size of 2^32 - 4 or more: the format says to store 2^32 - 1
      if (v != 5 && x.vsize > 0xFFFFFFFCll) o.u32(0xFFFFFFFFu);
      else o.count(v, x.vsize);
      if (v == 1) o.u32((uint32_t)x.begin); else o.i64(x.begin);
    }
  }
}

// Number of elements and bytes of a fixed-size variable from its dimensions.
bool var_extent(const Header &h, const Var &v, int64_t *nelems, std::string *err) {
  int64_t n = 1;
  for (int64_t id : v.dimids) {
    if (id < 0 || id >= (int64_t)h.dims.size()) { *err = "variable '" + v.name + "' names a missing dimension"; return false; }
    const int64_t len = h.dims[(size_t)id].len;
    if (len == 0) { *err = "variable '" + v.name + "' is a record variable (not supported)"; return false; }
    if (len < 0 || (n != 0 && len > INT64_MAX / 8 / n)) { *err = "variable '" + v.name + "' is too large"; return false; }
    n *= len;
  }
  *nelems = n;
  return true;
}

void swap32(uint32_t *p, size_t n) {
  for (size_t i = 0; i < n; i++) p[i] = __builtin_bswap32(p[i]);
}

}  // namespace

int type_size(int t) {
  switch (t) {
    case NC_BYTE: case NC_CHAR: case NC_UBYTE: return 1;
    case NC_SHORT: case NC_USHORT: return 2;
    case NC_INT: case NC_FLOAT: case NC_UINT: return 4;
    case NC_DOUBLE: case NC_INT64: case NC_UINT64: return 8;
    default: return 0;
  }
}

bool Att::as_int(int *out) const {
  if (nelems < 1) return false;
  const unsigned char *b = raw.data();
  switch (type) {
    case NC_BYTE: *out = (signed char)b[0]; return true;
    case NC_UBYTE: *out = b[0]; return true;
    case NC_SHORT: *out = (int16_t)(b[0] << 8 | b[1]); return true;
    case NC_USHORT: *out = b[0] << 8 | b[1]; return true;
    case NC_INT: case NC_UINT:
      *out = (int)((uint32_t)b[0] << 24 | (uint32_t)b[1] << 16 | (uint32_t)b[2] << 8 | (uint32_t)b[3]);
      return true;
    case NC_INT64: case NC_UINT64: {
      uint64_t v = 0;
      for (int i = 0; i < 8; i++) v = v << 8 | b[i];
      *out = (int)v;
      return true;
    }
    default: return false;  // text and floating-point attributes are not integers
  }
}

std::string Att::as_text() const {
  std::string s(raw.begin(), raw.end());
  while (!s.empty() && s.back() == '\0') s.pop_back();
  return s;
}

const Att *Header::find_att(const std::string &name) const {
  for (const Att &a : gatts) if (a.name == name) return &a;
  return nullptr;
}
const Var *Header::find_var(const std::string &name) const {
  for (const Var &v : vars) if (v.name == name) return &v;
  return nullptr;
}
int Header::find_dim(const std::string &name) const {
  for (size_t i = 0; i < dims.size(); i++) if (dims[i].name == name) return (int)i;
  return -1;
}

bool read_header(const std::string &path, Header *h, std::string *err) {
  *h = Header();
  const int fd = open(path.c_str(), O_RDONLY);
  if (fd < 0) { *err = sys_error("cannot open", path); return false; }
  struct stat st;
  if (fstat(fd, &st) != 0) { *err = sys_error("cannot stat", path); close(fd); return false; }
  In in(fd, (int64_t)st.st_size);
  bool ok = false;
  do {
    unsigned char magic[4];
    if (!in.bytes(magic, 4) || magic[0] != 'C' || magic[1] != 'D' || magic[2] != 'F') {
      *err = "'" + path + "' is not a NetCDF classic / CDF-5 file";
      if (!in.fail() && magic[0] == 0x89 && magic[1] == 'H' && magic[2] == 'D') *err += " (it is HDF5 / NetCDF-4)";
      break;
    }
    const int v = magic[3];
    if (v != 1 && v != 2 && v != 5) { *err = "unsupported CDF version " + std::to_string(v); break; }
    h->version = v;
    h->numrecs = in.count(v);
    const uint32_t dtag = in.u32();
    const int64_t nd = in.count(v);
    if (in.fail() || (dtag != kTagDim && !(dtag == 0 && nd == 0)) || nd < 0 || nd > 65536) { *err = "bad dimension list"; break; }
    bool bad = false;
    for (int64_t i = 0; i < nd && !bad; i++) {
      Dim d;
      if (!in.name(v, &d.name)) { bad = true; break; }
      d.len = in.count(v);
      if (in.fail() || d.len < 0) bad = true;
      h->dims.push_back(d);
    }
    if (bad) { *err = "truncated dimension list"; break; }
    if (!read_atts(in, v, &h->gatts, err)) break;
    const uint32_t vtag = in.u32();
    const int64_t nv = in.count(v);
    if (in.fail() || (vtag != kTagVar && !(vtag == 0 && nv == 0)) || nv < 0 || nv > 65536) { *err = "bad variable list"; break; }
    for (int64_t i = 0; i < nv && !bad; i++) {
      Var x;
      if (!in.name(v, &x.name)) { bad = true; break; }
      const int64_t ndims = in.count(v);
      if (in.fail() || ndims < 0 || ndims > 1024) { bad = true; break; }
      for (int64_t j = 0; j < ndims; j++) x.dimids.push_back(in.count(v));
      if (!read_atts(in, v, &x.atts, err)) { bad = true; break; }
      x.type = (int)in.u32();
      x.vsize = in.count(v);
      x.begin = v == 1 ? (int64_t)in.u32() : in.i64();
      if (in.fail() || type_size(x.type) == 0 || x.begin < 0) { bad = true; break; }
      h->vars.push_back(x);
    }
    if (bad) { if (err->empty()) *err = "truncated variable list"; break; }
    ok = true;
  } while (false);
  close(fd);
  return ok;
}

bool read_rows(const std::string &path, const Header &h, const std::string &var, int64_t start0,
               int64_t count0, void *out, std::string *err) {
  const Var *v = h.find_var(var);
  if (!v) { *err = "no variable '" + var + "' in '" + path + "'"; return false; }
  if (type_size(v->type) != 4 || v->dimids.size() != 2) { *err = "variable '" + var + "' is not a two-dimensional int / float array"; return false; }
  int64_t nelems = 0;
  if (!var_extent(h, *v, &nelems, err)) return false;
  const int64_t d0 = h.dims[(size_t)v->dimids[0]].len, d1 = h.dims[(size_t)v->dimids[1]].len;
  if (start0 < 0 || count0 < 0 || start0 + count0 > d0) {
    *err = "rows [" + std::to_string(start0) + ", " + std::to_string(start0 + count0) + ") of '" + var + "' lie outside its " + std::to_string(d0) + " rows";
    return false;
  }
  if (count0 == 0 || d1 == 0) return true;
  const int fd = open(path.c_str(), O_RDONLY);
  if (fd < 0) { *err = sys_error("cannot open", path); return false; }
  const int64_t bytes = count0 * d1 * 4;
  const int64_t off = v->begin + start0 * d1 * 4;
  char *p = static_cast<char *>(out);
  int64_t done = 0;
  bool ok = true;
  while (done < bytes) {
    const ssize_t r = pread(fd, p + done, (size_t)std::min<int64_t>(bytes - done, 1 << 30), (off_t)(off + done));
    if (r < 0) { *err = sys_error("read error in", path); ok = false; break; }
    if (r == 0) { *err = "'" + path + "' ends inside variable '" + var + "'"; ok = false; break; }
    done += r;
  }
  close(fd);
  if (ok) swap32(static_cast<uint32_t *>(out), (size_t)(count0 * d1));
  return ok;
}

// ---------------------------------------------------------------- Writer

Writer::~Writer() {
  if (fd_ >= 0) ::close(fd_);
}

int Writer::def_dim(const std::string &name, int64_t len) {
  Dim d;
  d.name = name;
  d.len = len;
  h_.dims.push_back(d);
  return (int)h_.dims.size() - 1;
}

int Writer::def_var(const std::string &name, int type, const std::vector<int> &dimids) {
  Var v;
  v.name = name;
  v.type = type;
  for (int id : dimids) v.dimids.push_back(id);
  h_.vars.push_back(v);
  return (int)h_.vars.size() - 1;
}

void Writer::put_att_text(const std::string &name, const std::string &text, bool with_nul) {
  Att a;
  a.name = name;
  a.type = NC_CHAR;
  a.raw.assign(text.begin(), text.end());
  if (with_nul) a.raw.push_back(0);
  a.nelems = (int64_t)a.raw.size();
  h_.gatts.push_back(a);
}

void Writer::put_att_int(const std::string &name, int value) {
  Att a;
  a.name = name;
  a.type = NC_INT;
  a.nelems = 1;
  const uint32_t u = (uint32_t)value;
  a.raw = {(unsigned char)(u >> 24), (unsigned char)(u >> 16), (unsigned char)(u >> 8), (unsigned char)u};
  h_.gatts.push_back(a);
}

bool Writer::begin_data(const std::string &path, std::string *err) {
  std::vector<int64_t> bytes(h_.vars.size());
  for (size_t i = 0; i < h_.vars.size(); i++) {
    int64_t n = 0;
    if (!var_extent(h_, h_.vars[i], &n, err)) return false;
    bytes[i] = pad4(n * type_size(h_.vars[i].type));
    h_.vars[i].vsize = bytes[i];
  }
  // Lay the variables out for a given version and say whether its fields can
  // hold the result.
  auto layout = [&](int version) -> bool {
    h_.version = version;
    for (Var &v : h_.vars) v.begin = 0;
    Out probe;
    encode_header(h_, probe);  // the header's size does not depend on the values of begin
    int64_t at = ((int64_t)probe.buf.size() + kDataAlign - 1) / kDataAlign * kDataAlign;
    bool fits = true;
    for (size_t i = 0; i < h_.vars.size(); i++) {
      h_.vars[i].begin = at;
      if (version == 1 && at > 0x7FFFFFFFll) fits = false;
      // CDF-1/2: only the last fixed-size variable may exceed 2^32 - 4 bytes
      if (version != 5 && bytes[i] > 0xFFFFFFFCll && i + 1 < h_.vars.size()) fits = false;
      at += bytes[i];
    }
    if (version != 5)
      for (const Dim &d : h_.dims) if (d.len > 0x7FFFFFFFll) fits = false;
    return fits;
  };
  int version = want_version_;
  if (version == 0) version = layout(1) ? 1 : 5;
  if (version != 1 && version != 2 && version != 5) { *err = "unsupported CDF version " + std::to_string(version); return false; }
  if (!layout(version)) { *err = "the data does not fit a CDF-" + std::to_string(version) + " file; use CDF-5"; return false; }
  Out o;
  encode_header(h_, o);
  fd_ = open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
  if (fd_ < 0) { *err = sys_error("cannot create", path); return false; }
  size_t done = 0;
  while (done < o.buf.size()) {
    const ssize_t w = pwrite(fd_, o.buf.data() + done, o.buf.size() - done, (off_t)done);
    if (w <= 0) { *err = sys_error("write error in", path); return false; }
    done += (size_t)w;
  }
  // size the file now, so that rows can be written in any order and unwritten
  // parts read as zeros
  int64_t end = (int64_t)o.buf.size();
  if (!h_.vars.empty()) end = std::max(end, h_.vars.back().begin + bytes.back());
  if (ftruncate(fd_, (off_t)end) != 0) { *err = sys_error("cannot size", path); return false; }
  return true;
}

bool Writer::write_rows(int varid, int64_t start0, int64_t count0, const void *data, std::string *err) {
  if (fd_ < 0) { *err = "write_rows before begin_data"; return false; }
  if (varid < 0 || varid >= (int)h_.vars.size()) { *err = "bad variable id"; return false; }
  const Var &v = h_.vars[(size_t)varid];
  if (type_size(v.type) != 4 || v.dimids.size() != 2) { *err = "variable '" + v.name + "' is not a two-dimensional int / float array"; return false; }
  const int64_t d0 = h_.dims[(size_t)v.dimids[0]].len, d1 = h_.dims[(size_t)v.dimids[1]].len;
  if (start0 < 0 || count0 < 0 || start0 + count0 > d0) { *err = "rows outside variable '" + v.name + "'"; return false; }
  const uint32_t *src = static_cast<const uint32_t *>(data);
  const int64_t n = count0 * d1;
  std::vector<uint32_t> tmp((size_t)std::min<int64_t>(n, 1 << 20));
  int64_t done = 0;
  while (done < n) {
    const int64_t m = std::min<int64_t>(n - done, (int64_t)tmp.size());
    for (int64_t i = 0; i < m; i++) tmp[(size_t)i] = __builtin_bswap32(src[done + i]);
    int64_t w = 0;
    while (w < m * 4) {
      const ssize_t r = pwrite(fd_, reinterpret_cast<const char *>(tmp.data()) + w, (size_t)(m * 4 - w),
                               (off_t)(v.begin + (start0 * d1 + done) * 4 + w));
      if (r <= 0) { *err = std::string("write error: ") + strerror(errno); return false; }
      w += r;
    }
    done += m;
  }
  return true;
}

bool Writer::close(std::string *err) {
  if (fd_ < 0) return true;
  const int rc = ::close(fd_);
  fd_ = -1;
  if (rc != 0) { *err = std::string("close failed: ") + strerror(errno); return false; }
  return true;
}

}  // namespace ncfile
