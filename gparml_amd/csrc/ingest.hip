// Shard ingest: the reference re-parses each shard's CSV with numpy.genfromtxt in every mapper
// call (local_MapReduce.py:197, 325; ~3 us per number, i.e. ~300 s for a 1e6 x 100 shard).  Here the file is parsed once,
// natively and in parallel, and Y then stays resident in HBM.  Host code only (no device work in this file).
#include "gp_common.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace gp {
namespace {

struct Mapped {
  const char* p = nullptr;
  size_t n = 0;
  int fd = -1;
  ~Mapped() {
    if (p && n) munmap(const_cast<char*>(p), n);
    if (fd >= 0) close(fd);
  }
};

int map_file(const char* path, Mapped& m) {
  m.fd = open(path, O_RDONLY);
  if (m.fd < 0) return fail(nullptr, GP_ERR_BAD_ARG, "cannot open %s: %s", path, strerror(errno));
  struct stat st;
  if (fstat(m.fd, &st) != 0) return fail(nullptr, GP_ERR_BAD_ARG, "cannot stat %s: %s", path, strerror(errno));
  m.n = (size_t)st.st_size;
  if (m.n == 0) return GP_OK;
  void* p = mmap(nullptr, m.n, PROT_READ, MAP_PRIVATE, m.fd, 0);
  if (p == MAP_FAILED) { m.n = 0; return fail(nullptr, GP_ERR_BAD_ARG, "cannot map %s: %s", path, strerror(errno)); }
  m.p = static_cast<const char*>(p);
  return GP_OK;
}

// a data line: not empty / whitespace only, and not a '#' comment (numpy.genfromtxt defaults)
bool is_data_line(const char* b, const char* e) {
  while (b < e && (*b == ' ' || *b == '\t' || *b == '\r')) ++b;
  return b < e && *b != '#';
}

// start offsets of all data lines
void index_lines(const Mapped& m, std::vector<size_t>& starts) {
  const char* p = m.p;
  const char* end = m.p + m.n;
  while (p < end) {
    const char* nl = static_cast<const char*>(memchr(p, '\n', (size_t)(end - p)));
    const char* le = nl ? nl : end;
    if (is_data_line(p, le)) starts.push_back((size_t)(p - m.p));
    p = le + 1;
  }
}

int count_fields(const char* b, const char* e) {
  int c = 1;
  for (; b < e; ++b) c += (*b == ',');
  return c;
}

const char* line_end(const Mapped& m, size_t start) {
  const char* b = m.p + start;
  const char* nl = static_cast<const char*>(memchr(b, '\n', m.n - start));
  const char* e = nl ? nl : m.p + m.n;
  while (e > b && (e[-1] == '\r' || e[-1] == ' ' || e[-1] == '\t')) --e;
  return e;
}

// one field [b, e): a number in any strtod format; empty or unparsable -> NaN (genfromtxt's missing value)
double parse_field(const char* b, const char* e) {
  while (b < e && (*b == ' ' || *b == '\t')) ++b;
  while (e > b && (e[-1] == ' ' || e[-1] == '\t')) --e;
  const size_t len = (size_t)(e - b);
  if (len == 0 || len > 63) return NAN;
  char buf[64];
  memcpy(buf, b, len);
  buf[len] = 0;
  char* endp = nullptr;
  const double v = strtod(buf, &endp);
  return (endp == buf + len) ? v : NAN;
}

}  // namespace
}  // namespace gp

using namespace gp;

extern "C" int gp_csv_shape(const char* path, int64_t* rows, int64_t* cols) {
  if (!path || !rows || !cols) return GP_ERR_BAD_ARG;
  Mapped m;
  int rc = map_file(path, m);
  if (rc != GP_OK) return rc;
  std::vector<size_t> starts;
  index_lines(m, starts);
  *rows = (int64_t)starts.size();
  *cols = starts.empty() ? 0 : count_fields(m.p + starts[0], line_end(m, starts[0]));
  return GP_OK;
}

extern "C" int gp_csv_read(const char* path, double* out, int64_t rows, int64_t cols, int threads) {
  if (!path || !out || rows < 0 || cols < 0) return GP_ERR_BAD_ARG;
  Mapped m;
  int rc = map_file(path, m);
  if (rc != GP_OK) return rc;
  std::vector<size_t> starts;
  index_lines(m, starts);
  if ((int64_t)starts.size() != rows) return fail(nullptr, GP_ERR_BAD_ARG, "%s has %zu data lines, expected %lld", path, starts.size(), (long long)rows);
  if (rows == 0) return GP_OK;
  int nt = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
  nt = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(nt, 64), rows / 1024 + 1));
  std::vector<int64_t> bad(nt, -1);
  auto work = [&](int t) {
    const int64_t r0 = rows * t / nt, r1 = rows * (t + 1) / nt;
    for (int64_t r = r0; r < r1; ++r) {
      const char* b = m.p + starts[r];
      const char* e = line_end(m, starts[r]);
      double* dst = out + r * cols;
      int64_t c = 0;
      while (true) {
        const char* comma = static_cast<const char*>(memchr(b, ',', (size_t)(e - b)));
        const char* fe = comma ? comma : e;
        if (c < cols) dst[c] = parse_field(b, fe);
        ++c;
        if (!comma) break;
        b = comma + 1;
      }
      if (c != cols && bad[t] < 0) bad[t] = r;
    }
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < nt; ++t) pool.emplace_back(work, t);
  work(0);
  for (auto& th : pool) th.join();
  for (int t = 0; t < nt; ++t)
    if (bad[t] >= 0) return fail(nullptr, GP_ERR_BAD_ARG, "%s: data line %lld does not have %lld columns", path, (long long)bad[t] + 1, (long long)cols);
  return GP_OK;
}
