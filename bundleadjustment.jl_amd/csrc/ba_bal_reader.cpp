// BAL reader: readfile(filename, T)  (reference: src/ReadFiles.jl:9-53).
//
// File format (https://grail.cs.washington.edu/projects/bal/): line 1 "ncams npnts nobs"; nobs lines
// "cam pnt x y" (0-based indices); 9*ncams lines of camera parameters in the order r(3) t(3) f k1 k2;
// 3*npnts lines of point coordinates.  Plain text or bzip2 (.bz2).  libbz2 has no development header in the
// image, so the three stream calls used are declared here and resolved from the runtime libbz2.so.1.0 with
// dlopen (the reference uses CodecBzip2, src/ReadFiles.jl:2,11).
#include <dlfcn.h>

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ba_hip.h"

void ba_set_error(const char *fmt, ...);

namespace {

typedef void *(*bz_open_t)(const char *, const char *);
typedef int (*bz_read_t)(void *, void *, int);
typedef void (*bz_close_t)(void *);

bool ends_with(const std::string &s, const char *suf) {
  size_t n = strlen(suf);
  return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

int slurp(const char *path, std::vector<char> &buf) {
  std::string sp(path);
  if (ends_with(sp, ".bz2")) {
    void *h = dlopen("libbz2.so.1.0", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("libbz2.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) {
      ba_set_error("cannot load libbz2 runtime: %s", dlerror());
      return BA_ERR_IO;
    }
    bz_open_t bzopen = (bz_open_t)dlsym(h, "BZ2_bzopen");
    bz_read_t bzread = (bz_read_t)dlsym(h, "BZ2_bzread");
    bz_close_t bzclose = (bz_close_t)dlsym(h, "BZ2_bzclose");
    if (!bzopen || !bzread || !bzclose) {
      ba_set_error("libbz2 lacks BZ2_bzopen/bzread/bzclose");
      return BA_ERR_IO;
    }
    void *f = bzopen(path, "rb");
    if (!f) {
      ba_set_error("cannot open %s", path);
      return BA_ERR_IO;
    }
    const int CH = 1 << 22;
    size_t used = 0;
    for (;;) {
      if (buf.size() < used + CH) buf.resize(buf.size() * 2 + CH);
      int n = bzread(f, buf.data() + used, CH);
      if (n < 0) {
        bzclose(f);
        ba_set_error("bzip2 stream error in %s", path);
        return BA_ERR_IO;
      }
      if (n == 0) break;
      used += (size_t)n;
    }
    bzclose(f);
    buf.resize(used + 1);
    buf[used] = 0;
    return BA_OK;
  }
  FILE *f = fopen(path, "rb");
  if (!f) {
    ba_set_error("cannot open %s: %s", path, strerror(errno));
    return BA_ERR_IO;
  }
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  buf.resize((size_t)sz + 1);
  size_t got = fread(buf.data(), 1, (size_t)sz, f);
  fclose(f);
  if (got != (size_t)sz) {
    ba_set_error("short read on %s", path);
    return BA_ERR_IO;
  }
  buf[(size_t)sz] = 0;
  return BA_OK;
}

struct Cursor {
  const char *p;
  const char *end;
  bool ok = true;
  long long next_int() {
    char *e;
    errno = 0;
    long long v = strtoll(p, &e, 10);
    if (e == p || e > end) ok = false;
    p = e;
    return v;
  }
  // parse(T, str): one correctly rounded conversion per type (strtod / strtof)
  double next_f64() {
    char *e;
    double v = strtod(p, &e);
    if (e == p || e > end) ok = false;
    p = e;
    return v;
  }
  float next_f32() {
    char *e;
    float v = strtof(p, &e);
    if (e == p || e > end) ok = false;
    p = e;
    return v;
  }
};

template <typename T>
T next_real(Cursor &c);
template <>
double next_real<double>(Cursor &c) { return c.next_f64(); }
template <>
float next_real<float>(Cursor &c) { return c.next_f32(); }

template <typename T>
int read_body(const char *path, int64_t ncams, int64_t npnts, int64_t nobs, int64_t *cam_idx1, int64_t *pnt_idx1,
              T *pt2d, T *x0) {
  if (!path || !cam_idx1 || !pnt_idx1 || !pt2d || !x0) {
    ba_set_error("ba_read_bal: null argument");
    return BA_ERR_ARG;
  }
  std::vector<char> buf;
  int rc = slurp(path, buf);
  if (rc != BA_OK) return rc;
  Cursor c{buf.data(), buf.data() + buf.size() - 1};
  long long hc = c.next_int(), hp = c.next_int(), ho = c.next_int();
  if (!c.ok || hc != ncams || hp != npnts || ho != nobs) {
    ba_set_error("%s: header (%lld %lld %lld) does not match the sizes passed (%lld %lld %lld)", path, hc, hp, ho,
                 (long long)ncams, (long long)npnts, (long long)nobs);
    return BA_ERR_IO;
  }
  for (int64_t i = 0; i < nobs; i++) {  // ReadFiles.jl:21-27 (indices made 1-based)
    cam_idx1[i] = c.next_int() + 1;
    pnt_idx1[i] = c.next_int() + 1;
    pt2d[2 * i] = next_real<T>(c);
    pt2d[2 * i + 1] = next_real<T>(c);
  }
  for (int64_t i = 0; i < ncams; i++) {  // ReadFiles.jl:32-43: file order r t f k1 k2 -> stored r t k1 k2 f
    T *C = x0 + 3 * npnts + 9 * i;
    for (int j = 0; j < 6; j++) C[j] = next_real<T>(c);
    C[8] = next_real<T>(c);
    C[6] = next_real<T>(c);
    C[7] = next_real<T>(c);
  }
  for (int64_t k = 0; k < 3 * npnts; k++) x0[k] = next_real<T>(c);  // ReadFiles.jl:45-47
  if (!c.ok) {
    ba_set_error("%s: truncated or malformed BAL file", path);
    return BA_ERR_IO;
  }
  return BA_OK;
}

}  // namespace

extern "C" int ba_read_bal_header(const char *path, int64_t *ncams, int64_t *npnts, int64_t *nobs) {
  if (!path || !ncams || !npnts || !nobs) {
    ba_set_error("ba_read_bal_header: null argument");
    return BA_ERR_ARG;
  }
  // The header is the first line; for .bz2 the stream has to be opened anyway, so decode only a first chunk.
  std::string sp(path);
  char head[256];
  memset(head, 0, sizeof head);
  if (ends_with(sp, ".bz2")) {
    void *h = dlopen("libbz2.so.1.0", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("libbz2.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) {
      ba_set_error("cannot load libbz2 runtime: %s", dlerror());
      return BA_ERR_IO;
    }
    bz_open_t bzopen = (bz_open_t)dlsym(h, "BZ2_bzopen");
    bz_read_t bzread = (bz_read_t)dlsym(h, "BZ2_bzread");
    bz_close_t bzclose = (bz_close_t)dlsym(h, "BZ2_bzclose");
    void *f = (bzopen && bzread && bzclose) ? bzopen(path, "rb") : nullptr;
    if (!f) {
      ba_set_error("cannot open %s", path);
      return BA_ERR_IO;
    }
    int n = bzread(f, head, sizeof head - 1);
    bzclose(f);
    if (n <= 0) {
      ba_set_error("bzip2 stream error in %s", path);
      return BA_ERR_IO;
    }
  } else {
    FILE *f = fopen(path, "rb");
    if (!f) {
      ba_set_error("cannot open %s: %s", path, strerror(errno));
      return BA_ERR_IO;
    }
    size_t n = fread(head, 1, sizeof head - 1, f);
    fclose(f);
    if (n == 0) {
      ba_set_error("%s is empty", path);
      return BA_ERR_IO;
    }
  }
  long long a, b, c;
  if (sscanf(head, "%lld %lld %lld", &a, &b, &c) != 3 || a < 0 || b < 0 || c < 0) {
    ba_set_error("%s: bad BAL header", path);
    return BA_ERR_IO;
  }
  *ncams = a;
  *npnts = b;
  *nobs = c;
  return BA_OK;
}

extern "C" int ba_read_bal(const char *path, int64_t ncams, int64_t npnts, int64_t nobs, int64_t *cam_idx1,
                           int64_t *pnt_idx1, double *pt2d, double *x0) {
  return read_body<double>(path, ncams, npnts, nobs, cam_idx1, pnt_idx1, pt2d, x0);
}

extern "C" int ba_read_bal_f32(const char *path, int64_t ncams, int64_t npnts, int64_t nobs, int64_t *cam_idx1,
                               int64_t *pnt_idx1, float *pt2d, float *x0) {
  return read_body<float>(path, ncams, npnts, nobs, cam_idx1, pnt_idx1, pt2d, x0);
}
