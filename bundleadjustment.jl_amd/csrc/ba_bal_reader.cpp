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

// libbz2 entry points, resolved once
struct Bz2Api {
  bz_open_t open = nullptr;
  bz_read_t read = nullptr;
  bz_close_t close = nullptr;
};
int bz2_api(Bz2Api *api) {
  void *h = dlopen("libbz2.so.1.0", RTLD_NOW | RTLD_LOCAL);
  if (!h) h = dlopen("libbz2.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!h) {
    ba_set_error("cannot load libbz2 runtime: %s", dlerror());
    return BA_ERR_IO;
  }
  api->open = (bz_open_t)dlsym(h, "BZ2_bzopen");
  api->read = (bz_read_t)dlsym(h, "BZ2_bzread");
  api->close = (bz_close_t)dlsym(h, "BZ2_bzclose");
  if (!api->open || !api->read || !api->close) {
    ba_set_error("libbz2 lacks BZ2_bzopen/bzread/bzclose");
    return BA_ERR_IO;
  }
  return BA_OK;
}

// Streaming text source: the file (plain or bzip2) is decoded chunk by chunk into one fixed window and parsed as it
// arrives -- nothing of the size of the file is ever held (Final-13682 is 1.9 GB of text).  A token (a number of at most
// a few dozen characters) never straddles a refill: the unread tail is moved to the front of the window first.
struct Stream {
  static constexpr size_t GUARD = 256;  // refill when fewer than GUARD unread bytes remain
  size_t WIN = 8u << 20;                // window; BA_READER_WINDOW (bytes, >= 4096) overrides it (tests use a tiny one)
  std::vector<char> buf;
  size_t pos = 0, len = 0;
  bool eof = false, failed = false;
  FILE *fp = nullptr;
  void *bz = nullptr;
  Bz2Api api;
  std::string path;

  explicit Stream(size_t window = 0) {
    if (window) WIN = window;
    else if (const char *e = getenv("BA_READER_WINDOW")) {
      const long v = atol(e);
      if (v >= 4096) WIN = (size_t)v;
    }
  }
  int open(const char *p) {
    path = p;
    buf.resize(WIN + 1);
    if (ends_with(path, ".bz2")) {
      int rc = bz2_api(&api);
      if (rc != BA_OK) return rc;
      bz = api.open(p, "rb");
      if (!bz) {
        ba_set_error("cannot open %s", p);
        return BA_ERR_IO;
      }
    } else {
      fp = fopen(p, "rb");
      if (!fp) {
        ba_set_error("cannot open %s: %s", p, strerror(errno));
        return BA_ERR_IO;
      }
    }
    refill();
    return failed ? BA_ERR_IO : BA_OK;
  }
  void refill() {
    if (eof) return;
    memmove(buf.data(), buf.data() + pos, len - pos);
    len -= pos;
    pos = 0;
    while (len < WIN && !eof) {
      long n;
      if (bz) {
        const size_t want = WIN - len < (size_t)(1 << 30) ? WIN - len : (size_t)(1 << 30);
        n = api.read(bz, buf.data() + len, (int)want);
        if (n < 0) {
          ba_set_error("bzip2 stream error in %s", path.c_str());
          failed = true;
          eof = true;
          break;
        }
      } else {
        n = (long)fread(buf.data() + len, 1, WIN - len, fp);
        if (n == 0 && ferror(fp)) {
          ba_set_error("read error on %s", path.c_str());
          failed = true;
        }
      }
      if (n == 0) eof = true;
      len += (size_t)n;
    }
    buf[len] = 0;  // strtod / strtoll stop here at the latest
  }
  inline const char *cur() {
    if (len - pos < GUARD && !eof) refill();
    return buf.data() + pos;
  }
  inline void advance_to(const char *e) { pos = (size_t)(e - buf.data()); }
  void close() {
    if (bz) api.close(bz);
    if (fp) fclose(fp);
    bz = nullptr;
    fp = nullptr;
  }
  ~Stream() { close(); }
};

struct Cursor {
  Stream *s;
  bool ok = true;
  long long next_int() {
    const char *p = s->cur();
    char *e;
    errno = 0;
    long long v = strtoll(p, &e, 10);
    if (e == p) ok = false;
    s->advance_to(e);
    return v;
  }
  // parse(T, str): one correctly rounded conversion per type (strtod / strtof)
  double next_f64() {
    const char *p = s->cur();
    char *e;
    double v = strtod(p, &e);
    if (e == p) ok = false;
    s->advance_to(e);
    return v;
  }
  float next_f32() {
    const char *p = s->cur();
    char *e;
    float v = strtof(p, &e);
    if (e == p) ok = false;
    s->advance_to(e);
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
  Stream in;
  int rc = in.open(path);
  if (rc != BA_OK) return rc;
  Cursor c{&in};
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
  if (in.failed) return BA_ERR_IO;  // message set by the stream
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
  // The header is the first line: only the first window of the stream is decoded (one bzip2 block), not the file.
  char head[256];
  memset(head, 0, sizeof head);
  {
    Stream in(4096);
    int rc = in.open(path);
    if (rc != BA_OK) return rc;
    if (in.len == 0) {
      ba_set_error("%s is empty", path);
      return BA_ERR_IO;
    }
    memcpy(head, in.buf.data(), in.len < sizeof head - 1 ? in.len : sizeof head - 1);
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
