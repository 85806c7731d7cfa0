/*
 * jda_oracle.c -- CPU restatement of the JDA sliding-window detect path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this.  The shipped library (libjda.so)
 * never links, loads or calls anything in oracle/.
 *
 * What it restates (each function cites the reference lines it follows):
 *   dialect C   -- reference c/jda.c (fp32, truncating coordinates)
 *   dialect CPP -- reference src/jda/{cascador,cart,data,btcart}.cpp (fp64,
 *                  round() coordinates), similarity transform off (the shipped
 *                  configuration); detect method 1 (growing window) and method 0
 *                  (true image pyramid); half/quarter images and pyramid levels
 *                  through a restatement of cv::resize (see SURVEY.md 8c for why
 *                  all of dialect CPP is unpinned)
 *
 * Pinning: dialect C is checked bit-for-bit against the reference's own
 * c/jda.c compiled from /root/reference (oracle/_ref, see oracle/build.py)
 * by tests/test_oracle_vs_reference.py, and against the golden vectors under
 * tests/golden/ that were produced by that build.  Dialect CPP cannot be
 * compiled here (needs OpenCV/jsmnpp/liblinear): PARITY UNPINNED for it; it
 * is cross-checked against dialect C where the two must agree, and (r06)
 * against a second restatement made independently from the reference's C++
 * (oracle/cpp_reading2.py, plain Python; tests/test_cpp_second_reading.py:
 * bit for bit per window and per image on single-scale models, method 1).
 * That reading found one divergence, fixed here and in the product: Validate
 * honours the header's training status (cascador.cpp:177-209), dialect C does
 * not (c/jda.c:499-505).  Agreement of two readings narrows, it does not pin.
 *
 * Unlike the reference, cascade dimensions are run-time values, and every
 * window reports where and why the walk stopped (carts evaluated, score,
 * leaf-path hash, shape) so reject decisions and tree paths can be compared
 * window by window.
 *
 * Build: gcc -std=c99 -O2 -ffp-contract=off (no -march=native, no
 * -ffast-math): an FMA contraction changes results.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  int scale, lm1, lm2, th;
  double off[4]; /* o1x o1y o2x o2y, as stored (f32 files widen exactly) */
} orc_node;

typedef struct {
  int T, K, L, D, node_n, leaf_n, dim;
  int real_bytes;
  int hdr_stage, hdr_cart;        /* header ints 5, 6: the training status (cascador.cpp:84-104, 136-141) */
  double *mean_shape;             /* [dim]                 */
  orc_node *nodes;                /* [T*K*node_n]          */
  double *leaf, *cth, *cmean, *cstd;
  double *w;                      /* [T][K*leaf_n][dim]    */
  /* fp32 views for dialect C: plain narrowing casts, c/jda.c:509-552 */
  float *mean_shape_f, *off_f, *leaf_f, *cth_f, *cmean_f, *cstd_f, *w_f;
} orc_model;

/* ------------------------------------------------------------------ model */

static long long orc_stream_bytes(int T, int K, int L, int D, int rb) {
  long long node_n = (1LL << (D - 1)) - 1, leaf_n = 1LL << (D - 1);
  long long cart = node_n * (16 + 4 * rb) + leaf_n * rb + 3 * rb;
  return 28 + 2LL * L * rb + (long long)T * (K * cart + (long long)K * leaf_n * 2 * L * rb) + 4;
}

static double rd_real(const unsigned char **p, int rb) {
  double v;
  if (rb == 8) { memcpy(&v, *p, 8); }
  else { float f; memcpy(&f, *p, 4); v = f; }
  *p += rb;
  return v;
}
static int rd_i32(const unsigned char **p) { int v; memcpy(&v, *p, 4); *p += 4; return v; }

void orc_free(orc_model *m) {
  if (!m) return;
  free(m->mean_shape); free(m->nodes); free(m->leaf); free(m->cth); free(m->cmean); free(m->cstd);
  free(m->w); free(m->mean_shape_f); free(m->off_f); free(m->leaf_f); free(m->cth_f);
  free(m->cmean_f); free(m->cstd_f); free(m->w_f); free(m);
}

/* Stream layout: reference README.md:84-111, cascador.cpp:79-164, cart.cpp:406-450. */
orc_model *orc_load(const char *path) {
  FILE *f = fopen(path, "rb");
  if (!f) return NULL;
  fseek(f, 0, SEEK_END);
  long long size = ftell(f);
  fseek(f, 0, SEEK_SET);
  unsigned char *buf = (unsigned char *)malloc(size > 0 ? size : 1);
  if (size < 32 || fread(buf, 1, size, f) != (size_t)size) { fclose(f); free(buf); return NULL; }
  fclose(f);
  const unsigned char *p = buf;
  orc_model *m = (orc_model *)calloc(1, sizeof(orc_model));
  (void)rd_i32(&p);
  m->T = rd_i32(&p); m->K = rd_i32(&p); m->L = rd_i32(&p); m->D = rd_i32(&p);
  m->hdr_stage = rd_i32(&p); m->hdr_cart = rd_i32(&p);
  if (m->T < 1 || m->T > 16 || m->K < 1 || m->L < 1 || m->D < 2 || m->D > 12) { free(buf); free(m); return NULL; }
  int rb = 0;
  if (size == orc_stream_bytes(m->T, m->K, m->L, m->D, 8)) rb = 8;
  else if (size == orc_stream_bytes(m->T, m->K, m->L, m->D, 4)) rb = 4;
  if (!rb) { free(buf); free(m); return NULL; }
  m->real_bytes = rb;
  m->node_n = (1 << (m->D - 1)) - 1; m->leaf_n = 1 << (m->D - 1); m->dim = 2 * m->L;
  long long carts = (long long)m->T * m->K;
  size_t wn = (size_t)carts * m->leaf_n * m->dim;
  m->mean_shape = (double *)malloc(sizeof(double) * m->dim);
  m->nodes = (orc_node *)malloc(sizeof(orc_node) * carts * m->node_n);
  m->leaf = (double *)malloc(sizeof(double) * carts * m->leaf_n);
  m->cth = (double *)malloc(sizeof(double) * carts);
  m->cmean = (double *)malloc(sizeof(double) * carts);
  m->cstd = (double *)malloc(sizeof(double) * carts);
  m->w = (double *)malloc(sizeof(double) * wn);
  for (int i = 0; i < m->dim; i++) m->mean_shape[i] = rd_real(&p, rb);
  for (int t = 0; t < m->T; t++) {
    for (int k = 0; k < m->K; k++) {
      long long c = (long long)t * m->K + k;
      for (int i = 0; i < m->node_n; i++) {
        orc_node *n = &m->nodes[c * m->node_n + i];
        n->scale = rd_i32(&p); n->lm1 = rd_i32(&p); n->lm2 = rd_i32(&p);
        for (int j = 0; j < 4; j++) n->off[j] = rd_real(&p, rb);
        n->th = rd_i32(&p);
      }
      for (int i = 0; i < m->leaf_n; i++) m->leaf[c * m->leaf_n + i] = rd_real(&p, rb);
      m->cth[c] = rd_real(&p, rb); m->cmean[c] = rd_real(&p, rb); m->cstd[c] = rd_real(&p, rb);
    }
    size_t per = (size_t)m->K * m->leaf_n * m->dim;
    for (size_t i = 0; i < per; i++) m->w[(size_t)t * per + i] = rd_real(&p, rb);
  }
  free(buf);
  /* fp32 views */
  m->mean_shape_f = (float *)malloc(sizeof(float) * m->dim);
  m->off_f = (float *)malloc(sizeof(float) * 4 * carts * m->node_n);
  m->leaf_f = (float *)malloc(sizeof(float) * carts * m->leaf_n);
  m->cth_f = (float *)malloc(sizeof(float) * carts);
  m->cmean_f = (float *)malloc(sizeof(float) * carts);
  m->cstd_f = (float *)malloc(sizeof(float) * carts);
  m->w_f = (float *)malloc(sizeof(float) * wn);
  for (int i = 0; i < m->dim; i++) m->mean_shape_f[i] = (float)m->mean_shape[i];
  for (long long i = 0; i < carts * m->node_n; i++)
    for (int j = 0; j < 4; j++) m->off_f[4 * i + j] = (float)m->nodes[i].off[j];
  for (long long i = 0; i < carts * m->leaf_n; i++) m->leaf_f[i] = (float)m->leaf[i];
  for (long long i = 0; i < carts; i++) {
    m->cth_f[i] = (float)m->cth[i]; m->cmean_f[i] = (float)m->cmean[i]; m->cstd_f[i] = (float)m->cstd[i];
  }
  for (size_t i = 0; i < wn; i++) m->w_f[i] = (float)m->w[i];
  return m;
}

void orc_dims(const orc_model *m, int *out6) {
  out6[0] = m->T; out6[1] = m->K; out6[2] = m->L; out6[3] = m->D; out6[4] = m->real_bytes; out6[5] = m->dim;
}

/* ---------------------------------------------------------------- resize */

/* Bilinear down-scale, reference c/jda.c:203-230: ratio=(src-1)/dst in float,
 * source index by truncation, four taps summed left to right, truncated to u8. */
void orc_resize(const unsigned char *src, int sw, int sh, unsigned char *dst, int dw, int dh) {
  const float rx = (float)(sw - 1) / dw;
  const float ry = (float)(sh - 1) / dh;
  for (int i = 0; i < dh; i++) {
    const float fy = ry * i;
    const int y0 = (int)fy;
    const float wy = fy - y0;
    for (int j = 0; j < dw; j++) {
      const float fx = rx * j;
      const int x0 = (int)fx;
      const float wx = fx - x0;
      const unsigned char *q = src + (size_t)y0 * sw + x0;
      const int p00 = q[0], p01 = q[1], p10 = q[sw], p11 = q[sw + 1];
      const float v = p00 * (1.f - wx) * (1.f - wy) + p01 * (wx) * (1.f - wy) +
                      p10 * (1.f - wx) * (wy) + p11 * (wx) * (wy);
      dst[(size_t)i * dw + j] = (unsigned char)v;
    }
  }
}

void orc_pyramid_dims(int w, int h, int *hw, int *hh, int *qw, int *qh) {
  const float r = 1.f / sqrtf(2.f); /* c/jda.c:450-456 */
  *hw = (int)(w * r); *hh = (int)(h * r);
  *qw = w / 2; *qh = h / 2;
}

/* ------------------------------------------------------- cv::resize (8UC1) */

/* Restatement of OpenCV's cv::resize(src, dst, Size(dw,dh)) for CV_8UC1 with the
 * default INTER_LINEAR, as dialect CPP uses it (reference cascador.cpp:300-303,
 * 329-331): fixed-point bilinear with 11-bit coefficients
 * (imgproc/src/imgwarp.cpp of the 2.4/3.x line the reference targets, README.md:30),
 * including the switch to the 2x2 box average when both scale factors are exactly 2.
 * PARITY UNPINNED: there is no OpenCV in this container to check it against. */
static int orc_cv_round(double v) { return (int)lrint(v); }          /* cvRound: round half to even */
static short orc_sat_short(float v) { int i = orc_cv_round(v); return (short)(i < -32768 ? -32768 : (i > 32767 ? 32767 : i)); }

void orc_resize_cv(const unsigned char *src, int sw, int sh, unsigned char *dst, int dw, int dh) {
  const double inv_sx = (double)dw / sw, inv_sy = (double)dh / sh;
  const double scale_x = 1. / inv_sx, scale_y = 1. / inv_sy;
  const int isx = (int)scale_x, isy = (int)scale_y;     /* saturate_cast<int>(double) rounds; exact for 2.0 */
  if (fabs(scale_x - 2.) < 2.220446049250313e-16 && fabs(scale_y - 2.) < 2.220446049250313e-16 && isx == 2 && isy == 2) {
    /* INTER_LINEAR with scale exactly 2 in both directions is routed to the fast INTER_AREA path */
    for (int y = 0; y < dh; y++)
      for (int x = 0; x < dw; x++) {
        const unsigned char *p = src + (size_t)(2 * y) * sw + 2 * x;
        dst[(size_t)y * dw + x] = (unsigned char)((p[0] + p[1] + p[sw] + p[sw + 1] + 2) >> 2);
      }
    return;
  }
  int *xofs = (int *)malloc(sizeof(int) * dw), *yofs = (int *)malloc(sizeof(int) * dh);
  short *ialpha = (short *)malloc(sizeof(short) * 2 * dw), *ibeta = (short *)malloc(sizeof(short) * 2 * dh);
  int xmax = dw;
  for (int dx = 0; dx < dw; dx++) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = (int)floorf(fx);
    fx -= sx;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx + 1 >= sw) { if (dx < xmax) xmax = dx; if (sx >= sw - 1) { fx = 0; sx = sw - 1; } }
    xofs[dx] = sx;
    ialpha[2 * dx] = orc_sat_short((1.f - fx) * 2048.f);
    ialpha[2 * dx + 1] = orc_sat_short(fx * 2048.f);
  }
  for (int dy = 0; dy < dh; dy++) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = (int)floorf(fy);
    fy -= sy;
    yofs[dy] = sy;
    ibeta[2 * dy] = orc_sat_short((1.f - fy) * 2048.f);
    ibeta[2 * dy + 1] = orc_sat_short(fy * 2048.f);
  }
  int *row0 = (int *)malloc(sizeof(int) * dw), *row1 = (int *)malloc(sizeof(int) * dw);
  for (int dy = 0; dy < dh; dy++) {
    int sy0 = yofs[dy], sy1 = yofs[dy] + 1;
    if (sy0 < 0) sy0 = 0;
    if (sy0 > sh - 1) sy0 = sh - 1;
    if (sy1 < 0) sy1 = 0;
    if (sy1 > sh - 1) sy1 = sh - 1;
    const unsigned char *S0 = src + (size_t)sy0 * sw, *S1 = src + (size_t)sy1 * sw;
    for (int dx = 0; dx < dw; dx++) {
      const int sx = xofs[dx];
      if (dx < xmax) {
        row0[dx] = S0[sx] * ialpha[2 * dx] + S0[sx + 1] * ialpha[2 * dx + 1];
        row1[dx] = S1[sx] * ialpha[2 * dx] + S1[sx + 1] * ialpha[2 * dx + 1];
      } else {
        row0[dx] = S0[sx] * 2048;
        row1[dx] = S1[sx] * 2048;
      }
    }
    const int b0 = ibeta[2 * dy], b1 = ibeta[2 * dy + 1];
    for (int dx = 0; dx < dw; dx++)
      dst[(size_t)dy * dw + dx] = (unsigned char)((((b0 * (row0[dx] >> 4)) >> 16) + ((b1 * (row1[dx] >> 4)) >> 16) + 2) >> 2);
  }
  free(xofs); free(yofs); free(ialpha); free(ibeta); free(row0); free(row1);
}

/* ------------------------------------------------------------ enumeration */

typedef struct { int win, step, nx, ny; long long base; } orc_level;

/* Window sizes and steps of reference c/jda.c:320-339 (+459-460).
 * Returns the level count (<=0: the reference loop would not terminate). */
static int orc_levels_c(int w, int h, float scale, int min_size, int max_size,
                        orc_level *lv, int cap, long long *total) {
  if (min_size < 24) min_size = 24;
  if (max_size <= 0) max_size = w < h ? w : h;
  if (max_size > w) max_size = w;
  if (max_size > h) max_size = h;
  int win = 24, n = 0;
  long long tot = 0;
  if ((int)(24 * scale) <= 24) return -1;
  while (win < min_size) win = (int)(win * scale);
  for (; win <= max_size; win = (int)(win * scale)) {
    if (n >= cap) return -2;
    lv[n].win = win;
    lv[n].step = (int)(win * 0.1f);
    lv[n].nx = (w - win) / lv[n].step + 1;
    lv[n].ny = (h - win) / lv[n].step + 1;
    lv[n].base = tot;
    tot += (long long)lv[n].nx * lv[n].ny;
    n++;
  }
  *total = tot;
  return n;
}

long long orc_count_windows_c(int w, int h, float scale, int min_size, int max_size, int *n_levels) {
  orc_level lv[256];
  long long tot = 0;
  int n = orc_levels_c(w, h, scale, min_size, max_size, lv, 256, &tot);
  if (n_levels) *n_levels = n;
  return n < 0 ? -1 : tot;
}

/* ------------------------------------------------------- dialect C window */

#define FNV_SEED 2166136261u
#define FNV_STEP(h, v) (((h) ^ (unsigned)(v)) * 16777619u)

typedef struct {
  const unsigned char *data; /* top-left of the image this scale reads */
  int w, h;                  /* image bounds (for the guarded read)    */
  int ox, oy;                /* window origin inside that image        */
} orc_view;

/* One window through the cascade, reference c/jda.c:357-414.
 * shape: in = unused, out = shape after the last completed stage.
 * Returns carts evaluated (rejecting cart included); *alive tells whether
 * every cart was passed.  The final threshold is NOT applied here. */
static int orc_walk_c(const orc_model *m, const orc_view *views, int win,
                      float *shape, int *lbf, float *score_out, unsigned *hash_out, int *alive) {
  const int dim = m->dim, node_n = m->node_n, leaf_n = m->leaf_n;
  float score = 0.f;
  unsigned hash = FNV_SEED;
  int evaluated = 0;
  *alive = 1;
  memcpy(shape, m->mean_shape_f, sizeof(float) * dim);
  for (int t = 0; t < m->T && *alive; t++) {
    for (int k = 0; k < m->K; k++) {
      const long long c = (long long)t * m->K + k;
      int at = 0;
      for (int d = 0; d < m->D - 1; d++) {
        const orc_node *nd = &m->nodes[c * node_n + at];
        const float *of = &m->off_f[4 * (c * node_n + at)];
        const orc_view *v = &views[nd->scale];
        /* landmark + offset, scaled by the window side, truncated toward 0,
         * clamped into the window (c/jda.c:373-389; note every scale uses
         * the full window side, c/jda.c:347-354) */
        const float ax = shape[2 * nd->lm1] + of[0];
        const float ay = shape[2 * nd->lm1 + 1] + of[1];
        const float bx = shape[2 * nd->lm2] + of[2];
        const float by = shape[2 * nd->lm2 + 1] + of[3];
        int ix1 = (int)(ax * win), iy1 = (int)(ay * win);
        int ix2 = (int)(bx * win), iy2 = (int)(by * win);
        if (ix1 < 0) ix1 = 0; else if (ix1 >= win) ix1 = win - 1;
        if (ix2 < 0) ix2 = 0; else if (ix2 >= win) ix2 = win - 1;
        if (iy1 < 0) iy1 = 0; else if (iy1 >= win) iy1 = win - 1;
        if (iy2 < 0) iy2 = 0; else if (iy2 >= win) iy2 = win - 1;
        /* Guarded read: for scale!=0 the reference indexes the half/quarter
         * image with full-window coordinates and can leave it (undefined
         * behaviour, SURVEY.md header item 6).  Rows/columns are clamped to
         * the image here; identical to the reference whenever it is in bounds. */
        int gx1 = v->ox + ix1, gy1 = v->oy + iy1, gx2 = v->ox + ix2, gy2 = v->oy + iy2;
        if (gx1 >= v->w) gx1 = v->w - 1;
        if (gy1 >= v->h) gy1 = v->h - 1;
        if (gx2 >= v->w) gx2 = v->w - 1;
        if (gy2 >= v->h) gy2 = v->h - 1;
        const int feat = (int)v->data[(size_t)gy1 * v->w + gx1] - (int)v->data[(size_t)gy2 * v->w + gx2];
        at = 2 * at + (feat <= nd->th ? 1 : 2);             /* c/jda.c:392-393 */
      }
      const int leaf = at - node_n;
      evaluated++;
      hash = FNV_STEP(hash, leaf);
      score += m->leaf_f[c * leaf_n + leaf];                 /* c/jda.c:396 */
      score = (score - m->cmean_f[c]) / m->cstd_f[c];         /* c/jda.c:397 */
      if (score < m->cth_f[c]) { *alive = 0; break; }         /* c/jda.c:399 */
      lbf[k] = k * leaf_n + leaf;
    }
    if (!*alive) break;
    /* stage regression: rows added in cart order, c/jda.c:404-411 */
    const float *ws = &m->w_f[(size_t)t * m->K * leaf_n * dim];
    for (int k = 0; k < m->K; k++) {
      const float *row = ws + (size_t)lbf[k] * dim;
      for (int i = 0; i < dim; i++) shape[i] += row[i];
    }
  }
  *score_out = score;
  *hash_out = hash;
  return evaluated;
}

typedef struct {
  unsigned char *half, *quarter;
  int hw, hh, qw, qh;
} orc_pyr;

static void orc_pyr_build(const unsigned char *img, int w, int h, orc_pyr *p) {
  orc_pyramid_dims(w, h, &p->hw, &p->hh, &p->qw, &p->qh);
  p->half = (unsigned char *)malloc((size_t)(p->hw > 0 ? p->hw : 1) * (p->hh > 0 ? p->hh : 1));
  p->quarter = (unsigned char *)malloc((size_t)(p->qw > 0 ? p->qw : 1) * (p->qh > 0 ? p->qh : 1));
  if (p->hw > 0 && p->hh > 0) orc_resize(img, w, h, p->half, p->hw, p->hh);
  if (p->qw > 0 && p->qh > 0) orc_resize(img, w, h, p->quarter, p->qw, p->qh);
}

static void orc_views(const unsigned char *img, int w, int h, const orc_pyr *p, int x, int y, orc_view *v) {
  const float r = 1.f / sqrtf(2.f);
  v[0].data = img; v[0].w = w; v[0].h = h; v[0].ox = x; v[0].oy = y;
  v[1].data = p->half; v[1].w = p->hw; v[1].h = p->hh; v[1].ox = (int)(x * r); v[1].oy = (int)(y * r);
  v[2].data = p->quarter; v[2].w = p->qw; v[2].h = p->qh; v[2].ox = x / 2; v[2].oy = y / 2;
}

/* Per-window trace in scan order. Any output may be NULL. Returns windows. */
long long orc_trace_c(const orc_model *m, const unsigned char *img, int w, int h,
                      float scale, int min_size, int max_size,
                      int *carts_n, float *score, unsigned *path_hash, float *shapes) {
  orc_level lv[256];
  long long tot = 0;
  const int nl = orc_levels_c(w, h, scale, min_size, max_size, lv, 256, &tot);
  if (nl < 0) return -1;
  orc_pyr pyr;
  orc_pyr_build(img, w, h, &pyr);
  float *shape = (float *)malloc(sizeof(float) * m->dim);
  int *lbf = (int *)malloc(sizeof(int) * m->K);
  long long id = 0;
  for (int l = 0; l < nl; l++) {
    for (int iy = 0; iy < lv[l].ny; iy++) {
      for (int ix = 0; ix < lv[l].nx; ix++, id++) {
        orc_view v[3];
        orc_views(img, w, h, &pyr, ix * lv[l].step, iy * lv[l].step, v);
        float s; unsigned hsh; int alive;
        const int n = orc_walk_c(m, v, lv[l].win, shape, lbf, &s, &hsh, &alive);
        if (carts_n) carts_n[id] = n;
        if (score) score[id] = s;
        if (path_hash) path_hash[id] = hsh;
        if (shapes) memcpy(shapes + (size_t)id * m->dim, shape, sizeof(float) * m->dim);
      }
    }
  }
  free(shape); free(lbf); free(pyr.half); free(pyr.quarter);
  return tot;
}

/* --------------------------------------------------------- dialect C NMS */

/* Reference c/jda.c:237-316. keep[i] = 1 for survivors (scan order kept). */
static void orc_nms_c(const int *bb, const float *sc, int n, float overlap, unsigned char *keep) {
  int *ord = (int *)malloc(sizeof(int) * (n > 0 ? n : 1));
  for (int i = 0; i < n; i++) { ord[i] = i; keep[i] = 1; }
  for (int i = 0; i + 1 < n; i++)
    for (int j = i + 1; j < n; j++)
      if (sc[ord[i]] < sc[ord[j]]) { int t = ord[i]; ord[i] = ord[j]; ord[j] = t; }
  for (int i = 0; i + 1 < n; i++) {
    const int a = ord[i];
    if (!keep[a]) continue;
    for (int j = i + 1; j < n; j++) {
      const int b = ord[j];
      if (!keep[b]) continue;
      const int ax = bb[3 * a], ay = bb[3 * a + 1], as = bb[3 * a + 2];
      const int bx = bb[3 * b], by = bb[3 * b + 1], bs = bb[3 * b + 2];
      const int x1 = ax > bx ? ax : bx, y1 = ay > by ? ay : by;
      const int x2 = (ax + as < bx + bs) ? ax + as : bx + bs;
      const int y2 = (ay + as < by + bs) ? ay + as : by + bs;
      const int iw = x2 - x1 > 0 ? x2 - x1 : 0, ih = y2 - y1 > 0 ? y2 - y1 : 0;
      const float ov = (float)(iw * ih) / (float)(as * as + bs * bs - iw * ih);
      if (ov > overlap) keep[b] = 0;
    }
  }
  free(ord);
}

/* Whole dialect-C detect of one frame = reference jdaDetect (c/jda.c:443-480).
 * Output arrays must hold one entry per window (see orc_count_windows_c).
 * Returns detections written, or -1. */
int orc_detect_c(const orc_model *m, const unsigned char *img, int w, int h,
                 float scale, int min_size, int max_size, float th, int do_nms,
                 int *bboxes, float *scores, float *shapes) {
  orc_level lv[256];
  long long tot = 0;
  const int nl = orc_levels_c(w, h, scale, min_size, max_size, lv, 256, &tot);
  if (nl < 0) return -1;
  orc_pyr pyr;
  orc_pyr_build(img, w, h, &pyr);
  float *shape = (float *)malloc(sizeof(float) * m->dim);
  int *lbf = (int *)malloc(sizeof(int) * m->K);
  int n = 0;
  for (int l = 0; l < nl; l++) {
    for (int iy = 0; iy < lv[l].ny; iy++) {
      for (int ix = 0; ix < lv[l].nx; ix++) {
        orc_view v[3];
        const int x = ix * lv[l].step, y = iy * lv[l].step;
        orc_views(img, w, h, &pyr, x, y, v);
        float s; unsigned hsh; int alive;
        (void)orc_walk_c(m, v, lv[l].win, shape, lbf, &s, &hsh, &alive);
        if (!alive) continue;
        if (s < th) continue;                                /* c/jda.c:414 */
        bboxes[3 * n] = x; bboxes[3 * n + 1] = y; bboxes[3 * n + 2] = lv[l].win;
        scores[n] = s;
        memcpy(shapes + (size_t)n * m->dim, shape, sizeof(float) * m->dim);
        n++;
      }
    }
  }
  free(shape); free(lbf); free(pyr.half); free(pyr.quarter);
  if (do_nms) {
    unsigned char *keep = (unsigned char *)malloc(n > 0 ? n : 1);
    orc_nms_c(bboxes, scores, n, 0.3f, keep);
    int o = 0;
    for (int i = 0; i < n; i++) {
      if (!keep[i]) continue;
      if (o != i) {
        memmove(bboxes + 3 * o, bboxes + 3 * i, sizeof(int) * 3);
        scores[o] = scores[i];
        memmove(shapes + (size_t)o * m->dim, shapes + (size_t)i * m->dim, sizeof(float) * m->dim);
      }
      o++;
    }
    free(keep);
    n = o;
    /* relocation, c/jda.c:465-474: multiply, then add */
    for (int i = 0; i < n; i++) {
      const int x = bboxes[3 * i], y = bboxes[3 * i + 1], sz = bboxes[3 * i + 2];
      float *sh = shapes + (size_t)i * m->dim;
      for (int j = 0; j < m->L; j++) {
        sh[2 * j] = sh[2 * j] * sz + x;
        sh[2 * j + 1] = sh[2 * j + 1] * sz + y;
      }
    }
  }
  return n;
}

/* ---------------------------------------------------------- dialect CPP */

/* Window sizes of detectMultiScale1, reference cascador.cpp:310-376. */
static int orc_levels_cpp(int w, int h, int minimum_size, int step, double factor,
                          orc_level *lv, int cap, long long *total) {
  if (minimum_size < 1 || step < 1 || (int)(minimum_size * factor) <= minimum_size) return -1;
  int win = minimum_size, n = 0;
  long long tot = 0;
  while (win <= w && win <= h) {
    if (n >= cap) return -2;
    lv[n].win = win; lv[n].step = step;
    lv[n].nx = (w - win) / step + 1; lv[n].ny = (h - win) / step + 1;
    lv[n].base = tot;
    tot += (long long)lv[n].nx * lv[n].ny;
    n++;
    win = (int)(win * factor);
  }
  *total = tot;
  return n;
}

long long orc_count_windows_cpp(int w, int h, int minimum_size, int step, double factor, int *n_levels) {
  orc_level lv[256];
  long long tot = 0;
  int n = orc_levels_cpp(w, h, minimum_size, step, factor, lv, 256, &tot);
  if (n_levels) *n_levels = n;
  return n < 0 ? -1 : tot;
}

/* Validate (cascador.cpp:166-211) on the origin-scale patch only, with
 * shift_size = 0 (test.cpp:17,75) and the identity STParameter
 * (data.cpp:68-70).  Cart::Forward cart.cpp:392-404; feature data.cpp:18-58
 * (round half away from zero, clamp common.hpp:227-232); delta shape summed
 * from zero then added, btcart.cpp:407-424.  Apply() with the identity
 * parameter is written out (1*(1*x+0*y)) because it is not a no-op for -0. */
/* Similarity transform of Validate (data.cpp:64-126, data.hpp:18-50): sR that maps the
 * mean shape onto the current shape.  UNPINNED details restated from OpenCV: cv::norm of a
 * CV_64F row = sqrt of a sum of squares accumulated four at a time; `Mat_ /= double` is a
 * convertTo with alpha = 1./b, i.e. v*(1./b) + 0. */
typedef struct { double scale, r00, r01, r10, r11; } orc_stp;

static double orc_cvnorm(const double *v, int n) {
  double s = 0.;
  int i = 0;
  for (; i <= n - 4; i += 4) { const double v0 = v[i], v1 = v[i + 1], v2 = v[i + 2], v3 = v[i + 3]; s += v0 * v0 + v1 * v1 + v2 * v2 + v3 * v3; }
  for (; i < n; i++) s += v[i] * v[i];
  return sqrt(s);
}

static orc_stp orc_stp_calc(const double *s1, const double *s2, int L, int enabled, double *t1, double *t2) {
  orc_stp p = {1., 1., 0., 0., 1.};
  if (!enabled) return p;                                    /* data.cpp:68-70 */
  double x1c = 0., y1c = 0., x2c = 0., y2c = 0.;
  for (int i = 0; i < L; i++) { x1c += s1[2 * i]; y1c += s1[2 * i + 1]; x2c += s2[2 * i]; y2c += s2[2 * i + 1]; }
  x1c /= L; y1c /= L; x2c /= L; y2c /= L;
  for (int i = 0; i < L; i++) {
    t1[2 * i] = s1[2 * i] - x1c; t1[2 * i + 1] = s1[2 * i + 1] - y1c;
    t2[2 * i] = s2[2 * i] - x2c; t2[2 * i + 1] = s2[2 * i + 1] - y2c;
  }
  const double scale1 = orc_cvnorm(t1, 2 * L), scale2 = orc_cvnorm(t2, 2 * L);
  p.scale = scale1 / scale2;
  const double a1 = 1. / scale1, a2 = 1. / scale2;
  for (int i = 0; i < 2 * L; i++) { t1[i] = t1[i] * a1 + 0.; t2[i] = t2[i] * a2 + 0.; }
  double num = 0., den = 0.;
  for (int i = 0; i < L; i++) {
    num += t1[2 * i + 1] * t2[2 * i] - t1[2 * i] * t2[2 * i + 1];
    den += t1[2 * i] * t2[2 * i] + t1[2 * i + 1] * t2[2 * i + 1];
  }
  const double norm = sqrt(num * num + den * den);
  const double sn = num / norm, cs = den / norm;
  p.r00 = cs; p.r01 = -sn; p.r10 = sn; p.r11 = cs;
  return p;
}

static void orc_stp_apply(const orc_stp *p, double x, double y, double *x2, double *y2) {   /* data.hpp:42-45 */
  *x2 = p->scale * (p->r00 * x + p->r01 * y);
  *y2 = p->scale * (p->r10 * x + p->r11 * y);
}

static int g_orc_similarity = 0;   /* Config::with_similarity_transform (common.cpp:214) */
void orc_set_similarity_transform(int on) { g_orc_similarity = on ? 1 : 0; }

typedef struct {
  const unsigned char *data; int iw;   /* image and its row stride          */
  int ox, oy;                          /* patch origin inside that image    */
  int pw;                              /* patch width == height             */
} orc_patch;

static int orc_walk_cpp(const orc_model *m, const orc_patch *pt,
                        double *shape, double *delta, int *lbf, double *score_out, unsigned *hash_out, int *alive) {
  const int dim = m->dim, node_n = m->node_n, leaf_n = m->leaf_n;
  double score = 0.;
  unsigned hash = FNV_SEED;
  int n = 0;
  *alive = 1;
  for (int j = 0; j < m->L; j++) {            /* RandomShape with zero shift, data.cpp:225-236 */
    shape[2 * j] = m->mean_shape[2 * j] + 0.;
    shape[2 * j + 1] = m->mean_shape[2 * j + 1] + 0.;
  }
  double *tmp = (double *)malloc(sizeof(double) * 2 * dim);
  /* Validate runs stages [0, current_stage_idx) in full and then carts [0, current_cart_idx] of the stage in training
   * WITHOUT its regression (cascador.cpp:177-209); a finished trainer model carries (T, -1) (cascador.cpp:93-98), a float
   * file written by the C library (T + 1, -1) (c/jda.c:662-665): both run everything.  (Dialect C reads the two ints and
   * drops them, c/jda.c:499-505: it always runs T x K.)  Found by the second reading, oracle/cpp_reading2.py. */
  const int full = m->hdr_stage >= 0 && m->hdr_stage < m->T ? m->hdr_stage : m->T;
  const int part = full < m->T ? (m->hdr_cart + 1 < m->K ? m->hdr_cart + 1 : m->K) : 0;
  orc_stp stp_last; memset(&stp_last, 0, sizeof stp_last);
  stp_last = orc_stp_calc(shape, m->mean_shape, m->L, 0, tmp, tmp + dim);                 /* STParameter's default: the identity */
  for (int t = 0; t < full + (part > 0 ? 1 : 0); t++) {
    const int in_training = t == full;                                                    /* cascador.cpp:198-209 */
    const orc_stp stp = in_training ? stp_last : orc_stp_calc(shape, m->mean_shape, m->L, g_orc_similarity, tmp, tmp + dim);   /* cascador.cpp:180; the stage in training walks with the LAST stage's parameter */
    stp_last = stp;
    const int k_end = in_training ? part : m->K;
    for (int k = 0; k < k_end; k++) {
      const long long c = (long long)t * m->K + k;
      int at = 0; /* 0-based position in the stored node array == reference idx-1 */
      for (int d = 0; d < m->D - 1; d++) {
        const orc_node *nd = &m->nodes[c * node_n + at];
        const orc_patch *q = &pt[nd->scale];                    /* data.cpp:21-34 */
        const int width = q->pw, height = q->pw;                /* data.cpp:37-38: the PATCH's size */
        double o1x, o1y, o2x, o2y;
        orc_stp_apply(&stp, nd->off[0], nd->off[1], &o1x, &o1y);   /* data.cpp:33-34 */
        orc_stp_apply(&stp, nd->off[2], nd->off[3], &o2x, &o2y);
        const double x1 = (shape[2 * nd->lm1] + o1x) * width;
        const double y1 = (shape[2 * nd->lm1 + 1] + o1y) * height;
        const double x2 = (shape[2 * nd->lm2] + o2x) * width;
        const double y2 = (shape[2 * nd->lm2 + 1] + o2y) * height;
        int ix1 = (int)round(x1), iy1 = (int)round(y1), ix2 = (int)round(x2), iy2 = (int)round(y2);
        if (ix1 < 0) ix1 = 0;
        if (iy1 < 0) iy1 = 0;
        if (ix1 >= width) ix1 = width - 1;
        if (iy1 >= height) iy1 = height - 1;
        if (ix2 < 0) ix2 = 0;
        if (iy2 < 0) iy2 = 0;
        if (ix2 >= width) ix2 = width - 1;
        if (iy2 >= height) iy2 = height - 1;
        const int val = (int)q->data[(size_t)(q->oy + iy1) * q->iw + q->ox + ix1] -
                        (int)q->data[(size_t)(q->oy + iy2) * q->iw + q->ox + ix2];
        /* reference is 1-based: left child 2i, right 2i+1; 0-based: 2i+1 / 2i+2 */
        at = 2 * at + (val <= nd->th ? 1 : 2);
      }
      const int leaf = at - node_n;
      hash = FNV_STEP(hash, leaf);
      score += m->leaf[c * leaf_n + leaf];
      score = (score - m->cmean[c]) / m->cstd[c];
      n++;
      if (score < m->cth[c]) { *alive = 0; *score_out = score; *hash_out = hash; free(tmp); return n; }
      lbf[k] = k * leaf_n + leaf;
    }
    if (in_training) break;                  /* no global regression for the stage in training */
    const double *ws = &m->w[(size_t)t * m->K * leaf_n * dim];
    for (int i = 0; i < dim; i++) delta[i] = 0.;
    for (int k = 0; k < m->K; k++) {
      const double *row = ws + (size_t)lbf[k] * dim;
      for (int i = 0; i < dim; i++) delta[i] += row[i];
    }
    for (int j = 0; j < m->L; j++) {           /* stp_mc.Apply(delta, delta), btcart.cpp:422 */
      const double dx = delta[2 * j], dy = delta[2 * j + 1];
      orc_stp_apply(&stp, dx, dy, &delta[2 * j], &delta[2 * j + 1]);
    }
    for (int i = 0; i < dim; i++) shape[i] += delta[i];
  }
  free(tmp);
  *score_out = score;
  *hash_out = hash;
  return n;
}

/* The three images of detectMultiScale1 (cascador.cpp:319-331) and the three
 * patches of one window (cascador.cpp:340-353). */
typedef struct { unsigned char *half, *quarter; int hw, hh, qw, qh; } orc_pyr_cpp;

static void orc_pyr_cpp_build(const orc_model *m, const unsigned char *img, int w, int h, orc_pyr_cpp *p);
static int orc_has_multiscale(const orc_model *m);

static void orc_patches_cpp(const unsigned char *img, int w, const orc_pyr_cpp *p, int x, int y, int win, orc_patch *pt) {
  const double r = sqrt(2.);
  pt[0].data = img; pt[0].iw = w; pt[0].ox = x; pt[0].oy = y; pt[0].pw = win;
  pt[1].data = p->half; pt[1].iw = p->hw; pt[1].ox = (int)(x / r); pt[1].oy = (int)(y / r); pt[1].pw = (int)(win / r);
  pt[2].data = p->quarter; pt[2].iw = p->qw; pt[2].ox = x / 2; pt[2].oy = y / 2; pt[2].pw = win / 2;
}

static int orc_has_multiscale(const orc_model *m) {
  for (long long i = 0; i < (long long)m->T * m->K * m->node_n; i++)
    if (m->nodes[i].scale != 0) return 1;
  return 0;
}

void orc_resize_cv(const unsigned char *src, int sw, int sh, unsigned char *dst, int dw, int dh);

static void orc_pyr_cpp_build(const orc_model *m, const unsigned char *img, int w, int h, orc_pyr_cpp *p) {
  p->half = p->quarter = NULL;
  p->hw = (int)(w / sqrt(2.)); p->hh = (int)(h / sqrt(2.));       /* cascador.cpp:323-326 */
  p->qw = w / 2; p->qh = h / 2;
  if (!orc_has_multiscale(m)) return;                             /* the images are only read by scale!=0 nodes */
  p->half = (unsigned char *)malloc((size_t)(p->hw > 0 ? p->hw : 1) * (p->hh > 0 ? p->hh : 1));
  p->quarter = (unsigned char *)malloc((size_t)(p->qw > 0 ? p->qw : 1) * (p->qh > 0 ? p->qh : 1));
  if (p->hw > 0 && p->hh > 0) orc_resize_cv(img, w, h, p->half, p->hw, p->hh);
  if (p->qw > 0 && p->qh > 0) orc_resize_cv(img, w, h, p->quarter, p->qw, p->qh);
}

long long orc_trace_cpp(const orc_model *m, const unsigned char *img, int w, int h,
                        int minimum_size, int step, double factor,
                        int *carts_n, double *score, unsigned *path_hash, double *shapes) {
  orc_level lv[256];
  long long tot = 0;
  const int nl = orc_levels_cpp(w, h, minimum_size, step, factor, lv, 256, &tot);
  if (nl < 0) return -1;
  orc_pyr_cpp pyr;
  orc_pyr_cpp_build(m, img, w, h, &pyr);
  double *shape = (double *)malloc(sizeof(double) * m->dim);
  double *delta = (double *)malloc(sizeof(double) * m->dim);
  int *lbf = (int *)malloc(sizeof(int) * m->K);
  long long id = 0;
  for (int l = 0; l < nl; l++)
    for (int iy = 0; iy < lv[l].ny; iy++)
      for (int ix = 0; ix < lv[l].nx; ix++, id++) {
        double s; unsigned hsh; int alive;
        orc_patch pt[3];
        orc_patches_cpp(img, w, &pyr, ix * lv[l].step, iy * lv[l].step, lv[l].win, pt);
        const int n = orc_walk_cpp(m, pt, shape, delta, lbf, &s, &hsh, &alive);
        if (carts_n) carts_n[id] = n;
        if (score) score[id] = s;
        if (path_hash) path_hash[id] = hsh;
        if (shapes) memcpy(shapes + (size_t)id * m->dim, shape, sizeof(double) * m->dim);
      }
  free(shape); free(delta); free(lbf); free(pyr.half); free(pyr.quarter);
  return tot;
}

/* nms of cascador.cpp:387-429: ascending multimap (equal keys in insertion
 * order), pick the last, erase everything whose IoU with it exceeds overlap. */
static int orc_nms_cpp(const int *r, const double *sc, int n, double overlap, int *picked) {
  int *asc = (int *)malloc(sizeof(int) * (n > 0 ? n : 1));
  unsigned char *alive = (unsigned char *)malloc(n > 0 ? n : 1);
  for (int i = 0; i < n; i++) { asc[i] = i; alive[i] = 1; }
  for (int i = 1; i < n; i++) { /* stable insertion sort, ascending */
    int v = asc[i], j = i - 1;
    while (j >= 0 && sc[asc[j]] > sc[v]) { asc[j + 1] = asc[j]; j--; }
    asc[j + 1] = v;
  }
  int np = 0, hi = n - 1;
  for (;;) {
    while (hi >= 0 && !alive[hi]) hi--;
    if (hi < 0) break;
    const int last = asc[hi];
    picked[np++] = last;
    const double la = r[4 * last + 2] * r[4 * last + 3];
    for (int p = 0; p <= hi; p++) {
      if (!alive[p]) continue;
      const int idx = asc[p];
      const double x1 = r[4 * idx] > r[4 * last] ? r[4 * idx] : r[4 * last];
      const double y1 = r[4 * idx + 1] > r[4 * last + 1] ? r[4 * idx + 1] : r[4 * last + 1];
      const int ex = r[4 * idx] + r[4 * idx + 2], lx = r[4 * last] + r[4 * last + 2];
      const int ey = r[4 * idx + 1] + r[4 * idx + 3], ly = r[4 * last + 1] + r[4 * last + 3];
      const double x2 = ex < lx ? ex : lx, y2 = ey < ly ? ey : ly;
      const double ww = x2 - x1 > 0. ? x2 - x1 : 0., hh = y2 - y1 > 0. ? y2 - y1 : 0.;
      const double ia = r[4 * idx + 2] * r[4 * idx + 3];
      const double ov = ww * hh / (ia + la - ww * hh);
      if (ov > overlap) alive[p] = 0;
    }
    alive[hi] = 0; /* guards the reference's endless loop when overlap >= 1 */
  }
  free(asc); free(alive);
  return np;
}

static int orc_nms_cpp(const int *r, const double *sc, int n, double overlap, int *picked);

/* Picks, relocates and writes the final lists of Detect (cascador.cpp:443-476). */
static int orc_finish_cpp(const orc_model *m, const int *r0, const double *s0, const double *h0, int n,
                          double overlap, int do_nms, int *rects, double *scores, double *shapes) {
  int *picked = (int *)malloc(sizeof(int) * (n > 0 ? n : 1));
  int np;
  if (do_nms) np = orc_nms_cpp(r0, s0, n, overlap, picked);
  else { np = n; for (int i = 0; i < n; i++) picked[i] = i; }
  for (int i = 0; i < np; i++) {
    const int p = picked[i];
    memcpy(rects + 4 * i, r0 + 4 * p, sizeof(int) * 4);
    scores[i] = s0[p];
    const double *src = h0 + (size_t)p * m->dim;
    double *dst = shapes + (size_t)i * m->dim;
    for (int j = 0; j < m->L; j++) {           /* cascador.cpp:468-471 */
      dst[2 * j] = r0[4 * p] + src[2 * j] * r0[4 * p + 2];
      dst[2 * j + 1] = r0[4 * p + 1] + src[2 * j + 1] * r0[4 * p + 3];
    }
  }
  free(picked);
  return np;
}

/* JoinCascador::Detect with method 1 (cascador.cpp:431-477). Outputs sized per
 * window count. rects are (x,y,w,h). Returns detections. */
int orc_detect_cpp(const orc_model *m, const unsigned char *img, int w, int h,
                   int minimum_size, int step, double factor, double overlap, int do_nms,
                   int *rects, double *scores, double *shapes) {
  orc_level lv[256];
  long long tot = 0;
  const int nl = orc_levels_cpp(w, h, minimum_size, step, factor, lv, 256, &tot);
  if (nl < 0) return -1;
  orc_pyr_cpp pyr;
  orc_pyr_cpp_build(m, img, w, h, &pyr);
  double *shape = (double *)malloc(sizeof(double) * m->dim);
  double *delta = (double *)malloc(sizeof(double) * m->dim);
  int *lbf = (int *)malloc(sizeof(int) * m->K);
  int *r0 = (int *)malloc(sizeof(int) * 4 * (tot > 0 ? tot : 1));
  double *s0 = (double *)malloc(sizeof(double) * (tot > 0 ? tot : 1));
  double *h0 = (double *)malloc(sizeof(double) * m->dim * (tot > 0 ? tot : 1));
  int n = 0;
  for (int l = 0; l < nl; l++)
    for (int iy = 0; iy < lv[l].ny; iy++)
      for (int ix = 0; ix < lv[l].nx; ix++) {
        double s; unsigned hsh; int alive;
        const int x = ix * lv[l].step, y = iy * lv[l].step;
        orc_patch pt[3];
        orc_patches_cpp(img, w, &pyr, x, y, lv[l].win, pt);
        (void)orc_walk_cpp(m, pt, shape, delta, lbf, &s, &hsh, &alive);
        if (!alive) continue;
        r0[4 * n] = x; r0[4 * n + 1] = y; r0[4 * n + 2] = lv[l].win; r0[4 * n + 3] = lv[l].win;
        s0[n] = s;
        memcpy(h0 + (size_t)n * m->dim, shape, sizeof(double) * m->dim);
        n++;
      }
  const int np = orc_finish_cpp(m, r0, s0, h0, n, overlap, do_nms, rects, scores, shapes);
  free(shape); free(delta); free(lbf); free(r0); free(s0); free(h0); free(pyr.half); free(pyr.quarter);
  return np;
}

/* Method 0, the true image pyramid: detectMultiScale + detectSingleScale
 * (cascador.cpp:216-308).  A fixed origin_size x origin_size window slides with a
 * pixel step over an image that is shrunk by 1/factor per level with cv::resize
 * (cumulatively, cascador.cpp:300-303); the per-window cv::resize to
 * origin_size is the identity (same size); rects are scaled back with truncating
 * int *= double (cascador.cpp:290-295).  scale==0 models only (the per-window
 * half/quarter patches of this method are not reproduced).
 * max_windows bounds the output arrays; returns detections or -1. */
long long orc_count_windows_pyramid(int w, int h, int origin_size, int step, double factor, int *n_levels) {
  if (origin_size < 1 || step < 1 || !(factor > 1.0)) return -1;
  long long tot = 0;
  int nl = 0;
  while (w >= origin_size && h >= origin_size) {
    tot += (long long)((w - origin_size) / step + 1) * ((h - origin_size) / step + 1);
    nl++;
    w = (int)(w / factor); h = (int)(h / factor);
    if (nl > 4096) return -1;
  }
  if (n_levels) *n_levels = nl;
  return tot;
}

/* Method 0 with the config's three patch sizes (image_size.origin_size / half_size / quarter_size, common.cpp:129-131):
 * detectSingleScale resizes EVERY window's ROI to each of them (cascador.cpp:243-245) -- the first is the identity
 * (the window is origin_size wide), the other two feed the split nodes of scale 1 and 2 (data.cpp:21-34).
 * half_size = quarter_size = 0: the call of a single-scale model (multi-scale models are refused). */
int orc_detect_cpp_pyramid_ms(const orc_model *m, const unsigned char *img, int w, int h,
                              int origin_size, int half_size, int quarter_size, int step, double factor, double overlap,
                              int do_nms, int *rects, double *scores, double *shapes) {
  const int multi = orc_has_multiscale(m);
  if (multi && (half_size < 1 || quarter_size < 1)) return -1;
  int nl = 0;
  const long long tot = orc_count_windows_pyramid(w, h, origin_size, step, factor, &nl);
  if (tot < 0) return -1;
  double *shape = (double *)malloc(sizeof(double) * m->dim);
  double *delta = (double *)malloc(sizeof(double) * m->dim);
  int *lbf = (int *)malloc(sizeof(int) * m->K);
  int *r0 = (int *)malloc(sizeof(int) * 4 * (tot > 0 ? tot : 1));
  double *s0 = (double *)malloc(sizeof(double) * (tot > 0 ? tot : 1));
  double *h0 = (double *)malloc(sizeof(double) * m->dim * (tot > 0 ? tot : 1));
  unsigned char *cur = (unsigned char *)malloc((size_t)w * h);
  unsigned char *roi = (unsigned char *)malloc((size_t)origin_size * origin_size);
  unsigned char *ph = (unsigned char *)malloc((size_t)(half_size > 0 ? half_size * half_size : 1));
  unsigned char *pq = (unsigned char *)malloc((size_t)(quarter_size > 0 ? quarter_size * quarter_size : 1));
  memcpy(cur, img, (size_t)w * h);
  int width = w, height = h, n = 0;
  double scale = 1.;
  while (width >= origin_size && height >= origin_size) {
    for (int y = 0; y <= height - origin_size; y += step)
      for (int x = 0; x <= width - origin_size; x += step) {
        double s; unsigned hsh; int alive;
        orc_patch pt[3];
        pt[0].data = cur; pt[0].iw = width; pt[0].ox = x; pt[0].oy = y; pt[0].pw = origin_size;
        pt[1] = pt[0]; pt[2] = pt[0];
        if (multi) {
          /* cv::resize(img(roi), patch_h / patch_q, ...): the ROI is an image of its own to cv::resize */
          for (int r = 0; r < origin_size; r++) memcpy(roi + (size_t)r * origin_size, cur + (size_t)(y + r) * width + x, (size_t)origin_size);
          orc_resize_cv(roi, origin_size, origin_size, ph, half_size, half_size);
          orc_resize_cv(roi, origin_size, origin_size, pq, quarter_size, quarter_size);
          pt[1].data = ph; pt[1].iw = half_size; pt[1].ox = 0; pt[1].oy = 0; pt[1].pw = half_size;
          pt[2].data = pq; pt[2].iw = quarter_size; pt[2].ox = 0; pt[2].oy = 0; pt[2].pw = quarter_size;
        }
        (void)orc_walk_cpp(m, pt, shape, delta, lbf, &s, &hsh, &alive);
        if (!alive) continue;
        int rx = x, ry = y, rw = origin_size, rh = origin_size;
        rx *= scale; ry *= scale; rw *= scale; rh *= scale;      /* cascador.cpp:292-294 */
        r0[4 * n] = rx; r0[4 * n + 1] = ry; r0[4 * n + 2] = rw; r0[4 * n + 3] = rh;
        s0[n] = s;
        memcpy(h0 + (size_t)n * m->dim, shape, sizeof(double) * m->dim);
        n++;
      }
    scale *= factor;
    const int nw = (int)(width / factor), nh = (int)(height / factor);
    if (nw < 1 || nh < 1) break;
    unsigned char *nxt = (unsigned char *)malloc((size_t)nw * nh);
    orc_resize_cv(cur, width, height, nxt, nw, nh);
    free(cur); cur = nxt; width = nw; height = nh;
  }
  const int np = orc_finish_cpp(m, r0, s0, h0, n, overlap, do_nms, rects, scores, shapes);
  free(shape); free(delta); free(lbf); free(r0); free(s0); free(h0); free(cur); free(roi); free(ph); free(pq);
  return np;
}

int orc_detect_cpp_pyramid(const orc_model *m, const unsigned char *img, int w, int h,
                           int origin_size, int step, double factor, double overlap, int do_nms,
                           int *rects, double *scores, double *shapes) {
  if (orc_has_multiscale(m)) return -1;
  return orc_detect_cpp_pyramid_ms(m, img, w, h, origin_size, 0, 0, step, factor, overlap, do_nms, rects, scores, shapes);
}
