// libjda.so, host side: images of different sizes as ONE job (jdaDetectBatchRagged[Device], jdaDetectBatchCppRagged[Device];
// the reference's FDDB loop, src/test.cpp:100-170, is one Detect per image).
#include "detect.h"

namespace jda {

// ---------------------------------------------------------------- ragged batches (images of different sizes)
//
// The reference's FDDB loop calls Detect once per image (src/test.cpp:100-170), the C API once per jdaDetect; on a
// GPU that is one latency-bound pass per image.  A ragged job runs a list of differently sized images as a few
// passes: the window sizes of c/jda.c:331-333 are the same series for every image (an image uses the prefix that fits
// it), so the levels, their tile shapes and stage-0 tables are shared, and the images are staged with ONE row pitch.
// Per image the results are those of jdaDetect on that image.
// Dialect CPP (r06) -- the dialect the reference's fddb() itself runs, joincascador.Detect with method 1
// (src/test.cpp:142, cascador.cpp:310-376): its window sizes are minimum_size, int(win * factor), ... with a fixed pixel
// step (cascador.cpp:314,369), again one series for every image, of which an image takes the prefix that fits both its
// sides (cascador.cpp:333).  The same job structure serves it, on the fp64 instantiations of the kernels.

struct RaggedJob {
  int n = 0;
  const int* widths = nullptr; const int* heights = nullptr;
  const unsigned char* const* host_imgs = nullptr;     // tight images in host memory, or
  const uint8_t* d_base = nullptr; const size_t* d_offsets = nullptr;   // ... on the device at d_base + d_offsets[i]
  int pitch = 0;                    // common row pitch of the staged images (multiple of 16)
  ScanPlan levels;                  // global level list; nx, ny = nominal (mean) grids, width = pitch
  std::vector<int> n_lv;            // levels image i has (a prefix of the global list)
  // the window grid of every (image, level), counted ONCE (ragged_prepare): image i's levels at [geo_first[i], geo_first[i + 1]).
  // (r06: a 356-image shard's tables were 93 us of integer divisions on the host, in front of the GPU's first kernel --
  // the same quotients four times over; now 25)
  std::vector<uint32_t> geo_first;
  std::vector<uint16_t> gnx, gny;
  std::vector<long long> wins;      // candidate windows of image i
  PlanEntry* pe = nullptr;
  bool cpp = false;                 // dialect CPP (call: its parameters), else dialect C (scale, min_size, max_size)
  CppCall call{};
  float scale = 0.f; int min_size = 0, max_size = 0;
  int real_bytes() const { return cpp ? 8 : 4; }
  uint8_t* d_job_raw = nullptr;     // host job with a helper thread: every chunk's tight images go here ...
  std::vector<size_t> raw_off;      // ... chunk k at d_job_raw + raw_off[k]
};

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
// n / d for n < 65536 and 1 <= d < 65536 by one multiply: m = ceil(2^32 / d) overshoots n / d by less than 2^-16 < 1 / d.
struct Div16 {
  uint64_t m = 0; bool one = false;
  explicit Div16(int d = 1) { one = d <= 1; m = one ? 0 : (0xffffffffull / (uint64_t)d) + 1; }
  int operator()(int n) const { return one ? n : (int)(((uint64_t)(uint32_t)n * m) >> 32); }
};
static thread_local double t_prep_marks[3];       // debug_times = 2: ragged_prepare's pieces (levels planned, grids counted, plan fetched)

// Levels of image (w, h): the prefix of the job's global list whose windows fit (c/jda.c:321-322,332).
static int ragged_levels_of(const RaggedJob& job, int w, int h) {
  const int lim = std::min(w, h);
  int k = 0;
  while (k < (int)job.levels.levels.size() && job.levels.levels[k].win <= lim) k++;
  return k;
}

// Geometry of a ragged call: common pitch, global levels with nominal grids, the plan (tile shapes + tables).
// Returns 0 = ok, 1 = this job needs the per-image fallback, -1 = error.
static int ragged_prepare(Cascador* c, RaggedJob* job) {
  const float scale = job->scale; const int min_size = job->min_size, max_size = job->max_size;
  int max_w = 0, max_min = 0;
  for (int i = 0; i < job->n; i++) {
    if (job->widths[i] <= 0 || job->heights[i] <= 0) { fail("image " + std::to_string(i) + " has no pixels"); return -1; }
    if (job->widths[i] > 65535 || job->heights[i] > 65535) { fail("images wider or taller than 65535 pixels are not supported"); return -1; }
    max_w = std::max(max_w, job->widths[i]);
    max_min = std::max(max_min, std::min(job->widths[i], job->heights[i]));
  }
  std::string err;
  if (job->cpp ? !plan_dialect_cpp(max_min, max_min, job->call.minimum_size, job->call.step, job->call.factor, &job->levels, &err)
               : !plan_dialect_c(max_min, max_min, scale, min_size, max_size, &job->levels, &err)) { fail(err); return -1; }
  t_prep_marks[0] = now_ms();
  const int nl = (int)job->levels.levels.size();
  if (nl > kMaxLevels) return 1;
  int pitch = (max_w + 15) & ~15;
  if ((pitch & 255) == 0) pitch += 16;            // keep rows of neighbouring tiles off one memory channel
  job->pitch = pitch;
  job->levels.width = pitch; job->levels.height = max_min;
  // nominal grids: the mean over the images that have the level (tile shapes are chosen for them)
  std::vector<double> sx(nl, 0.0), sy(nl, 0.0);
  std::vector<long long> cnt(nl, 0);
  job->n_lv.resize(job->n);
  job->geo_first.resize((size_t)job->n + 1);
  job->wins.resize((size_t)job->n);
  {
    Div16 by_step[kMaxLevels];
    for (int l = 0; l < nl; l++) by_step[l] = Div16(job->levels.levels[l].step);
    size_t tot = 0;
    for (int i = 0; i < job->n; i++) { job->n_lv[i] = ragged_levels_of(*job, job->widths[i], job->heights[i]); job->geo_first[i] = (uint32_t)tot; tot += (size_t)job->n_lv[i]; }
    job->geo_first[job->n] = (uint32_t)tot;
    job->gnx.resize(tot); job->gny.resize(tot);
    for (int i = 0; i < job->n; i++) {
      const int k = job->n_lv[i], W = job->widths[i], H = job->heights[i];
      uint16_t* gx = job->gnx.data() + job->geo_first[i]; uint16_t* gy = job->gny.data() + job->geo_first[i];
      long long wi = 0;
      for (int l = 0; l < k; l++) {
        const Level& lv = job->levels.levels[l];
        const int nx = by_step[l](W - lv.win) + 1, ny = by_step[l](H - lv.win) + 1;      // c/jda.c:335-336
        gx[l] = (uint16_t)nx; gy[l] = (uint16_t)ny;
        sx[l] += nx; sy[l] += ny; cnt[l]++;
        wi += (long long)nx * ny;
      }
      job->wins[(size_t)i] = wi;
    }
  }
  t_prep_marks[1] = now_ms();
  if (c->kn.debug_times == 1) fprintf(stderr, "[jda] ragged prepare: levels of %d images counted\n", job->n);
  unsigned long long h = 1469598103934665603ull;
  for (int l = 0; l < nl; l++) {
    Level& lv = job->levels.levels[l];
    // quantised, so that jobs over similar image sets share a plan
    // (rounded UP: a nominal grid one window narrower than the images' cuts every row of tiles in two)
    const int qx = cnt[l] ? std::max(1, (int)std::ceil(sx[l] / (double)cnt[l] / 4.0) * 4) : 1;
    const int qy = cnt[l] ? std::max(1, (int)std::ceil(sy[l] / (double)cnt[l] / 4.0) * 4) : 1;
    lv.nx = qx; lv.ny = qy; lv.base = 0;
    h = (h ^ (unsigned long long)(qx * 65536 + qy)) * 1099511628211ull;
  }
  job->levels.windows = 0;
  unsigned sb; std::memcpy(&sb, &scale, 4);
  PlanKey key{pitch, nl, 3 /* ragged, dialect C */, (int)sb, std::max(min_size, 24), max_size <= 0 ? -1 : max_size, h};
  if (job->cpp) {
    unsigned long long fb; std::memcpy(&fb, &job->call.factor, 8);
    h = (h ^ fb) * 1099511628211ull;
    key = PlanKey{pitch, nl, 4 /* ragged, dialect CPP */, job->call.minimum_size, job->call.step, c->similarity, h};
  }
  std::unique_lock<std::mutex> lk(c->mu);
  if (!get_plan(c, lk, key, job->levels, job->cpp ? JDA_DIALECT_CPP : JDA_DIALECT_C, &job->pe, true)) return -1;      // (pinned; detect_ragged unpins)
  t_prep_marks[2] = now_ms();
  if (job->pe->dense_hint && !c->last_dense) job->pe->dense_hint = false;   // the per-image passes since then rejected most windows again
  if (!job->pe->fast_scan || job->pe->any_untiled || job->pe->dense_hint || c->kn.dense == 2) return 1;
  for (int l = 0; l < nl; l++) if (job->pe->hp.lv[l].tw * job->pe->hp.lv[l].th > 512) return 1;
  return 0;
}

// The level's tile re-cut for an image's own grid of nx x ny windows: as few tiles per row as the level's widest tile
// allows, evenly wide; the slack of a narrower tile goes into its height (up to 512 windows and the LDS the level's
// launch may use), again evenly.  An FDDB-sized image (77 x 64 windows of 46 pixels) gets 2 x 5 tiles of 39 x 13 windows
// (99 % of a 512-lane first phase) instead of 2 x 7 of 50 x 10 (60 %).
static void ragged_tile(const DevLevel& d, int nx, int ny, int th_lds, int* tw, int* th) {
  if (d.tiled == 2) th_lds = 512;                                  // global-pixel "tiles" are only window groups: no LDS limit
  const int tx = (nx + d.tw - 1) / d.tw;
  *tw = (nx + tx - 1) / tx;
  const int cap = std::max(1, std::min(th_lds, 512 / *tw));
  const int ty = (ny + cap - 1) / cap;
  *th = (ny + ty - 1) / ty;
}
// rows of windows a tile of level d may hold within lds_budget bytes of pixels
static int ragged_th_lds(const DevLevel& d, int pix_budget) {
  const int rows = pix_budget / std::max(1, d.pitch);
  return std::max(d.th, (rows - d.win) / std::max(1, d.step) + 1);
}

// Tables of images [i0, i0 + n) into the lane's pinned table buffer (and, for host images that do not lie back to
// back, the images into the lane's pinned staging buffer).  Two steps, so that the pass can put the images' repack on the
// stream before the (longer) block map is built: ragged_build_images -- the image records, the front of the table buffer
// [0, ch->images_bytes) -- and ragged_build_chunk, the rest.
static bool ragged_build_images(const RaggedJob& job, int i0, int n, Lane* ln, RaggedChunk* ch) {
  ch->i0 = i0; ch->n = n; ch->pitch = job.pitch;
  ch->widths = job.widths + i0; ch->heights = job.heights + i0;
  ch->host_imgs = job.host_imgs ? job.host_imgs + i0 : nullptr;
  ch->d_raw = job.d_base;
  const uint32_t geo0 = job.geo_first[i0];
  const int n_segs = (int)(job.geo_first[i0 + n] - geo0);
  // layout: image records first (what k_repack and k_post read), then segments and the block map; the block map's size is
  // only known after the tile cuts, so it comes last
  size_t o = 0;
  ch->off_rimg = o; o = align_up(o + (size_t)n * sizeof(RagImg), 256);
  ch->off_imgoff = o; o = align_up(o + (size_t)n * sizeof(unsigned long long), 256);
  ch->off_gidbase = o; o = align_up(o + ((size_t)n + 1) * sizeof(uint32_t), 256);      // (k_post: an image's gid range)
  ch->images_bytes = o;
  ch->off_segs = o; o = align_up(o + (size_t)n_segs * sizeof(RagSeg), 256);
  ch->off_blk = o;
  ch->n_segs = n_segs;
  ch->images_issued = false;
  {
    // (room for the block map as well, so that neither buffer moves between the two steps: about a tile per 200 windows)
    long long wsum = 0;
    for (int i = 0; i < n; i++) wsum += job.wins[(size_t)(i0 + i)];
    const size_t guess = o + ((size_t)(wsum / 96) + (size_t)n_segs * 2 + 1024) * sizeof(RagBlk);
    if (!ln->h_tab.reserve(guess) || !ln->rag_tab.reserve(guess)) return false;
  }
  uint8_t* tab = (uint8_t*)ln->h_tab.p;
  unsigned long long* img_off = (unsigned long long*)(tab + ch->off_imgoff);
  RagImg* rimg = (RagImg*)(tab + ch->off_rimg);
  ch->gid_base.assign(n + 1, 0);
  size_t dst = 0, src = 0;
  long long gid = 0;
  int max_h = 0;
  bool contiguous = job.host_imgs != nullptr;
  for (int i = 0; i < n; i++) {
    const int W = job.widths[i0 + i], H = job.heights[i0 + i];
    max_h = std::max(max_h, H);
    img_off[i] = dst;
    rimg[i].dst_off = dst; rimg[i].w = W; rimg[i].h = H;
    if (job.host_imgs) {
      if (!job.host_imgs[i0 + i]) { fail("null image pointer"); return false; }
      if (i > 0 && job.host_imgs[i0 + i] != job.host_imgs[i0 + i - 1] + (size_t)job.widths[i0 + i - 1] * job.heights[i0 + i - 1]) contiguous = false;
      rimg[i].src_off = src;                         // tight, back to back in the staging copy
      src += (size_t)W * H;
    } else {
      rimg[i].src_off = job.d_offsets[i0 + i];
    }
    dst += align_up((size_t)H * job.pitch, 256);
    ch->gid_base[i] = (uint32_t)gid;
    gid += job.wins[(size_t)(i0 + i)];
  }
  ch->gid_base[n] = (uint32_t)gid;
  std::memcpy(tab + ch->off_gidbase, ch->gid_base.data(), ((size_t)n + 1) * sizeof(uint32_t));
  if (gid > 0x7fffffffLL) { fail("ragged chunk has too many windows"); return false; }
  ch->windows = gid; ch->frame_bytes = dst + 256; ch->max_h = max_h;
  ch->raw_bytes = job.host_imgs ? src : 0;
  ch->host_contiguous = contiguous;
  if (!ln->rag_frames.reserve(ch->frame_bytes)) return false;
  if (job.host_imgs && !job.d_job_raw) {
    if (!ln->rag_raw.reserve(src + 16)) return false;
    if (!contiguous) {
      if (!ln->h_raw.reserve(src + 16)) return false;
      uint8_t* hr = (uint8_t*)ln->h_raw.p;
      for (int i = 0; i < n; i++) std::memcpy(hr + rimg[i].src_off, job.host_imgs[i0 + i], (size_t)rimg[i].w * rimg[i].h);
    }
  }
  return true;
}

static bool ragged_build_chunk(Cascador* c, const RaggedJob& job, int i0, int n, Lane* ln, RaggedChunk* ch) {
  const DevPlan& hp = job.pe->hp;
  const int nl = hp.n_levels;
  // ---- counts ----
  // How far a re-cut tile's pixels may outgrow the level's nominal tile (taller, narrower tiles for narrow images): as
  // far as the workgroups per CU stay what the nominal tile allows -- measured, a flat 1.4x took the 71/88-pixel levels
  // from 2 workgroups per CU to 1 and cost more than the fuller first phase gained.
  int th_lds[kMaxLevels];
  {
    const HostModel& hm = c->hm;
    const int rb = job.real_bytes();
    const int chunk = std::min(std::min(hm.K, (int)c->kn.handoff), scan_handoff_cap(hm.node_n(), hm.leaf_n(), rb));
    for (int l = 0; l < nl; l++) {
      const DevLevel& d = hp.lv[l];
      if (d.tiled == 2) { th_lds[l] = d.th; continue; }
      const int nominal = d.pitch * (d.win + (d.th - 1) * d.step);
      const int fixed = (int)scan_lds_bytes(0, chunk, hm.node_n(), hm.leaf_n(), rb, false, d.tw * d.th > 256 ? 512 : 256);
      // (workgroups per CU in LDS granules, host.h: a tile grown to 160 KB / 3 takes 43 granules and fits twice, not three times)
      const int per_cu = std::max(1, lds_wgs_per_cu(fixed + nominal));
      const int room = lds_bytes_for_wgs(per_cu) - fixed - 64;
      th_lds[l] = ragged_th_lds(d, std::max(nominal, std::min(room, nominal * (int)c->kn.ragged_tile_grow_pct / 100)));
    }
  }
  // ---- the tile cut of every (image, level) of the chunk, once: the level's tile re-cut for the image's grid.  The cut in x
  //      depends on (level, nx) only, the cut in y on (level, tw, ny): both memoised per chunk. ----
  struct Cut { uint16_t tw, th, tiles_x, tiles_y; };
  static thread_local std::vector<Cut> cuts;
  static thread_local std::vector<uint32_t> memo_x[kMaxLevels];        // [nx] -> tw | tiles_x << 16 (0: not yet)
  struct MemoY { uint32_t key; uint16_t th, tiles_y; };
  static thread_local MemoY memo_y[kMaxLevels][64];                    // direct-mapped on (tw, ny)
  const uint32_t geo0 = job.geo_first[i0];
  const int n_segs = (int)(job.geo_first[i0 + n] - geo0);
  cuts.resize((size_t)n_segs);
  for (int l = 0; l < nl; l++) { memo_x[l].clear(); for (auto& e : memo_y[l]) e.key = 0xffffffffu; }
  long long n_blk = 0;
  int th_max[kMaxLevels], win_max[kMaxLevels];       // rows / windows of the largest tile any image of the chunk cut from level l
  for (int l = 0; l < nl; l++) { th_max[l] = 1; win_max[l] = 1; }
  long long lds_blocks = 0;
  for (int i = 0; i < n; i++) {
    const uint32_t gi = job.geo_first[i0 + i];
    for (int l = 0; l < job.n_lv[i0 + i]; l++) {
      const DevLevel& d = hp.lv[l];
      const int nx = job.gnx[gi + l], ny = job.gny[gi + l];
      Cut& cu = cuts[gi - geo0 + l];
      if ((size_t)nx >= memo_x[l].size()) memo_x[l].resize((size_t)nx + 64, 0u);
      uint32_t mx = memo_x[l][nx];
      int tw, th, tiles_y;
      if (mx) tw = (int)(mx & 0xffffu);
      else {
        ragged_tile(d, nx, ny, th_lds[l], &tw, &th);
        mx = (uint32_t)tw | ((uint32_t)((nx + tw - 1) / tw) << 16);
        memo_x[l][nx] = mx;
      }
      const uint32_t ykey = ((uint32_t)tw << 16) | (uint32_t)ny;
      MemoY& my = memo_y[l][(ny * 7 + tw) & 63];
      if (my.key == ykey) { th = my.th; tiles_y = my.tiles_y; }
      else {
        int tw2;
        ragged_tile(d, nx, ny, th_lds[l], &tw2, &th);
        tiles_y = (ny + th - 1) / th;
        my.key = ykey; my.th = (uint16_t)th; my.tiles_y = (uint16_t)tiles_y;
      }
      cu.tw = (uint16_t)tw; cu.th = (uint16_t)th; cu.tiles_x = (uint16_t)(mx >> 16); cu.tiles_y = (uint16_t)tiles_y;
      const long long tiles = (long long)cu.tiles_x * tiles_y;
      n_blk += tiles;
      th_max[l] = std::max<int>(th_max[l], th); win_max[l] = std::max<int>(win_max[l], tw * th);
      if (d.tiled == 1) lds_blocks += tiles;
    }
  }
  if (n_blk > 0x7fffffffLL) { fail("ragged chunk has too many tiles"); return false; }
  ch->n_blk = (int)n_blk;
  ch->table_bytes = align_up(ch->off_blk + (size_t)n_blk * sizeof(RagBlk), 256);
  if (ch->table_bytes > ln->h_tab.bytes || ch->table_bytes > ln->rag_tab.bytes) {
    // (more tiles than ragged_build_images left room for: the buffers move.  What has been queued from / into them
    // finishes first, the pinned copy keeps its image records, and the pass uploads everything again)
    if (ch->images_issued) { if (hipStreamSynchronize(ln->stream) != hipSuccess) { fail("hipStreamSynchronize failed"); return false; } }
    ch->images_issued = false;
    if (!ln->h_tab.reserve(ch->table_bytes, ch->off_segs) || !ln->rag_tab.reserve(ch->table_bytes)) return false;
  }
  uint8_t* tab = (uint8_t*)ln->h_tab.p;
  RagSeg* segs = (RagSeg*)(tab + ch->off_segs);
  RagBlk* blk = (RagBlk*)(tab + ch->off_blk);
  const unsigned long long* img_off = (const unsigned long long*)(tab + ch->off_imgoff);
  // ---- segments ----
  std::vector<int> seg_first(n + 1, 0);
  int si = 0;
  for (int i = 0; i < n; i++) {
    long long gid = ch->gid_base[i];
    seg_first[i] = si;
    const uint32_t gi = job.geo_first[i0 + i];
    for (int l = 0; l < job.n_lv[i0 + i]; l++) {
      const DevLevel& d = hp.lv[l];
      const Cut& cu = cuts[gi - geo0 + l];
      RagSeg& sg = segs[si++];
      sg.img_off = img_off[i]; sg.gid_base = (uint32_t)gid;
      sg.nx = job.gnx[gi + l]; sg.ny = job.gny[gi + l];
      sg.tw = cu.tw; sg.th = cu.th;
      sg.tiles_x = cu.tiles_x;
      sg.level = (uint16_t)l; sg.image = (uint16_t)i; sg.pad0 = 0; sg.pad1 = 0;
      sg.win = d.win; sg.step = d.step; sg.pitch = d.pitch; sg.s0_table = d.s0_table; sg.tiled = d.tiled;
      sg.pad2 = sg.pad3 = sg.pad4 = 0;
      gid += (long long)sg.nx * sg.ny;
    }
  }
  seg_first[n] = si;
  // ---- block map and launches: one launch per LDS-tiled level (all of them in one when the chunk is small), one for
  //      the big-window LDS levels, one for the global-pixel levels.  Inside a launch the tiles of 8 images interleave,
  //      so that an image's tiles mostly land on one XCD's L2 (block b -> XCD b % 8). ----
  ch->launches.clear();
  int bi = 0;
  auto emit_group = [&](int l, int g0) {
    {
      int tiles[8], seg[8], most = 0;
      const int ge = std::min(n, g0 + 8);
      for (int i = g0; i < ge; i++) {
        tiles[i - g0] = 0; seg[i - g0] = -1;
        if (l < job.n_lv[i0 + i]) {
          const Cut& cu = cuts[seg_first[i] + l];
          tiles[i - g0] = (int)cu.tiles_x * cu.tiles_y;
          seg[i - g0] = seg_first[i] + l;
          most = std::max(most, tiles[i - g0]);
        }
      }
      for (int t = 0; t < most; t++)
        for (int j = 0; j < ge - g0; j++)
          if (t < tiles[j]) { blk[bi].seg = (uint32_t)seg[j]; blk[bi].tile = (uint32_t)t; bi++; }
    }
  };
  auto emit_level = [&](int l) { for (int g0 = 0; g0 < n; g0 += 8) emit_group(l, g0); };
  auto pix_of = [&](int l) { const DevLevel& d = hp.lv[l]; return d.pitch * (d.win + (th_max[l] - 1) * d.step); };
  const bool small = lds_blocks <= c->kn.merge_blocks;
  auto merged = [&](int mode) {
    RaggedChunk::Launch L{mode, 256, 0, bi, 0};
    // (group-major: the eight images of a group go through every level of the launch before the next group starts, so
    // that they stay in L2 from level to level -- level-major order cost the global-pixel launch 47 %)
    for (int g0 = 0; g0 < n; g0 += 8)
      for (int l = 0; l < nl; l++) if (hp.lv[l].tiled == mode) emit_group(l, g0);
    for (int l = 0; l < nl; l++) if (hp.lv[l].tiled == mode && mode != 2) L.pix_bytes = std::max(L.pix_bytes, pix_of(l));
    L.blk_n = bi - L.blk_base;
    if (L.blk_n > 0) ch->launches.push_back(L);
  };
  // ragged_merge: the LDS-tiled levels of a chunk as one launch per OCCUPANCY CLASS (levels whose workgroups fit the same
  // number of times into a CU's LDS) instead of one per level.  A chunk of 4 M windows over ten levels is ~900 workgroups
  // per level against 768 resident ones (3 per CU): every per-level launch runs a full round and a fifth of a second one
  // (r06 kernel trace of the dialect-CPP job: 60-85 us per launch where 35-40 would do).  Merged, a class is thousands of
  // workgroups and the tail is paid once.  -1 = auto: on for dialect CPP (no persistent scan there), off for dialect C
  // (its single-level launches are what k_scan_p takes).
  const long long merge_knob = c->kn.ragged_merge;
  const bool merge_classes = merge_knob < 0 ? job.cpp : merge_knob != 0;
  // Dialect C: the levels the persistent scan will take stay launches of their own; what it leaves to k_scan's closed
  // tiles (the 71- and 88-pixel levels: too few slots) shares launches per occupancy class like dialect CPP's levels do.
  bool own_launch[kMaxLevels];
  for (int l = 0; l < nl; l++) own_launch[l] = false;
  const bool merge_rest = !merge_classes && !job.cpp;        // (r06, a 356-image job: two closed-tile launches 176 -> one of 156 us; the 2,845-image job equal)
  if (merge_rest) {
    long long tiles_of[kMaxLevels];
    for (int l = 0; l < nl; l++) tiles_of[l] = 0;
    for (int i = 0; i < n; i++) {
      const uint32_t gi = job.geo_first[i0 + i] - geo0;
      for (int l = 0; l < job.n_lv[i0 + i]; l++) tiles_of[l] += (long long)cuts[gi + l].tiles_x * cuts[gi + l].tiles_y;
    }
    for (int l = 0; l < nl; l++) own_launch[l] = hp.lv[l].tiled == 1 && scan_p_takes_ragged(c, pix_of(l), tiles_of[l]);
  }
  if (small) merged(1);
  else if (merge_classes || merge_rest) {
    const HostModel& hm = c->hm;
    const int rb = job.real_bytes();
    const int chunk = std::min(std::min(hm.K, (int)c->kn.handoff), scan_handoff_cap(hm.node_n(), hm.leaf_n(), rb));
    int cls[kMaxLevels];
    bool any[kMaxLevels];
    for (int l = 0; l < nl; l++) {
      cls[l] = -1; any[l] = false;
      if (hp.lv[l].tiled != 1) continue;
      if (own_launch[l]) {
        RaggedChunk::Launch L{1, win_max[l] > 256 ? 512 : 256, pix_of(l), bi, 0};
        L.level = l;
        emit_level(l);
        L.blk_n = bi - L.blk_base;
        if (L.blk_n > 0) ch->launches.push_back(L);
        continue;
      }
      const int lds = (int)scan_lds_bytes(pix_of(l), chunk, hm.node_n(), hm.leaf_n(), rb, false, win_max[l] > 256 ? 512 : 256);
      cls[l] = std::max(1, std::min(8, lds_wgs_per_cu(lds)));
    }
    for (int i = 0; i < n; i++) for (int l = 0; l < job.n_lv[i0 + i]; l++) any[l] = true;
    for (int k = 8; k >= 1; k--) {
      RaggedChunk::Launch L{1, 256, 0, bi, 0};
      int members = 0, last = -1;
      for (int l = 0; l < nl; l++)
        if (cls[l] == k && any[l]) { members++; last = l; L.pix_bytes = std::max(L.pix_bytes, pix_of(l)); if (win_max[l] > 256) L.block = 512; }
      if (!members) continue;
      if (members == 1) { L.level = last; emit_level(last); }
      else
        for (int g0 = 0; g0 < n; g0 += 8)                   // (group-major, like merged(): eight images stay in L2 from level to level)
          for (int l = 0; l < nl; l++) if (cls[l] == k && any[l]) emit_group(l, g0);
      L.blk_n = bi - L.blk_base;
      if (L.blk_n > 0) ch->launches.push_back(L);
    }
  } else
    for (int l = 0; l < nl; l++)
      if (hp.lv[l].tiled == 1) {
        RaggedChunk::Launch L{1, win_max[l] > 256 ? 512 : 256, pix_of(l), bi, 0};
        L.level = l;
        emit_level(l);
        L.blk_n = bi - L.blk_base;
        if (L.blk_n > 0) ch->launches.push_back(L);
      }
  merged(3);
  merged(2);
  if (bi != ch->n_blk) { fail("internal: ragged block map size"); return false; }
  return true;
}

// NMS, relocation and the jdaResult of every image of a ragged chunk (dets sorted by gid = image, level, y, x).
// rows != nullptr (jdaDetectBatchRagged[Device]Rows): no jdaResult per image -- three allocations each, 1,065 for a
// 355-image job whose 1,170 detections are 270 KB -- but one row [frame_offset + image index, x, y, size, score, shape]
// per detection appended to *rows, images in order (what jdaResultsPack makes of the jdaResults, what the multi-GPU
// gather ships).
static double post_ragged(Cascador* c, const RaggedJob& job, const RaggedChunk& ch, const RawDets<float>& dets,
                          const jdaDetectOptions* opt, jdaResult* out, RowsOut<float>* rows = nullptr, int frame_offset = 0) {
  const double t0 = now_ms();
  const int L = c->hm.L, dim = c->hm.dim();
  const bool do_nms = !opt || opt->nms;
  const float overlap = opt ? opt->nms_overlap : 0.3f;
  const DevPlan& hp = job.pe->hp;
  std::vector<size_t> first(ch.n + 1, dets.gid.size());
  {
    size_t i = 0;
    for (int f = 0; f < ch.n; f++) {
      first[f] = i;
      while (i < dets.gid.size() && dets.gid[i] < ch.gid_base[f + 1]) i++;
    }
    first[ch.n] = i;
  }
  const bool some_posted = dets.p_n.size() == (size_t)ch.n;
  auto one = [&](int f, jdaResult& r) {
    if (some_posted && dets.p_n[f] >= 0) {           // post-processed on the device (k_post)
      const size_t k = (size_t)dets.p_n[f], r0 = (size_t)dets.p_first[f];
      r.n = (int)k; r.landmark_n = L;
      r.bboxes = (int*)std::malloc(std::max<size_t>(1, k * 3) * sizeof(int));
      r.scores = (float*)std::malloc(std::max<size_t>(1, k) * sizeof(float));
      r.shapes = (float*)std::malloc(std::max<size_t>(1, k * dim) * sizeof(float));
      if (k) {
        std::memcpy(r.bboxes, &dets.p_bb[r0 * 3], k * 3 * sizeof(int));
        std::memcpy(r.scores, &dets.p_sc[r0], k * sizeof(float));
        std::memcpy(r.shapes, &dets.p_sh[r0 * dim], k * dim * sizeof(float));
      }
      return;
    }
    const size_t a = first[f], cnt = first[f + 1] - a;
    static thread_local std::vector<int> bb, keep;
    bb.resize(cnt * 3);
    const int W = ch.widths[f], H = ch.heights[f];
    int l = 0;
    uint32_t lbase = ch.gid_base[f];
    int nx = 0, cntl = 0;
    auto level_grid = [&](int lv) {
      const DevLevel& d = hp.lv[lv];
      nx = (W - d.win) / d.step + 1;
      cntl = nx * ((H - d.win) / d.step + 1);
    };
    if (cnt) level_grid(0);
    for (size_t i = 0; i < cnt; i++) {          // gids ascend: levels are walked once
      const uint32_t g = dets.gid[a + i];
      while (g >= lbase + (uint32_t)cntl) { lbase += (uint32_t)cntl; l++; level_grid(l); }
      const uint32_t rel = g - lbase;
      const DevLevel& d = hp.lv[l];
      bb[3 * i] = (int)(rel % (uint32_t)nx) * d.step; bb[3 * i + 1] = (int)(rel / (uint32_t)nx) * d.step; bb[3 * i + 2] = d.win;
    }
    if (do_nms) nms_dialect_c_into(bb.data(), dets.score.data() + a, (int)cnt, overlap, &keep);
    else { keep.resize(cnt); std::iota(keep.begin(), keep.end(), 0); }
    r.n = (int)keep.size(); r.landmark_n = L;
    r.bboxes = (int*)std::malloc(std::max<size_t>(1, keep.size() * 3) * sizeof(int));
    r.scores = (float*)std::malloc(std::max<size_t>(1, keep.size()) * sizeof(float));
    r.shapes = (float*)std::malloc(std::max<size_t>(1, keep.size() * dim) * sizeof(float));
    for (size_t i = 0; i < keep.size(); i++) {
      const int k = keep[i];
      std::memcpy(r.bboxes + 3 * i, &bb[3 * k], 3 * sizeof(int));
      r.scores[i] = dets.score[a + k];
      float* sh = r.shapes + i * dim;
      std::memcpy(sh, &dets.shape[(a + k) * dim], dim * sizeof(float));
      relocate_dialect_c(sh, L, bb[3 * k], bb[3 * k + 1], bb[3 * k + 2]);
    }
  };
  if (rows) {
    const size_t rw = (size_t)5 + dim;
    for (int f = 0; f < ch.n; f++) {
      const float fr = (float)(frame_offset + ch.i0 + f);
      if (some_posted && dets.p_n[f] >= 0) {         // straight from what k_post left in pinned memory
        const size_t k = (size_t)dets.p_n[f], r0 = (size_t)dets.p_first[f];
        float* o = rows->grow(k * rw);
        for (size_t j = 0; j < k; j++, o += rw) {
          o[0] = fr; o[1] = (float)dets.p_bb[(r0 + j) * 3]; o[2] = (float)dets.p_bb[(r0 + j) * 3 + 1]; o[3] = (float)dets.p_bb[(r0 + j) * 3 + 2];
          o[4] = dets.p_sc[r0 + j];
          std::memcpy(o + 5, &dets.p_sh[(r0 + j) * dim], dim * sizeof(float));
        }
        continue;
      }
      jdaResult r{};
      one(f, r);
      float* o = rows->grow((size_t)r.n * rw);
      for (int j = 0; j < r.n; j++, o += rw) {
        o[0] = fr; o[1] = (float)r.bboxes[3 * j]; o[2] = (float)r.bboxes[3 * j + 1]; o[3] = (float)r.bboxes[3 * j + 2];
        o[4] = r.scores[j];
        std::memcpy(o + 5, r.shapes + (size_t)j * dim, dim * sizeof(float));
      }
      std::free(r.bboxes); std::free(r.scores); std::free(r.shapes);
    }
    return now_ms() - t0;
  }
  parallel_for(ch.n, [&](int f) { one(f, out[f]); }, dets.gid.size() < 6000);
  return now_ms() - t0;
}

// The same for dialect CPP: candidates of an image in scan order -> Rect(x, y, win, win) (cascador.cpp:339), NMS by score
// (cascador.cpp:387-429), relocation (462-474), jdaResultD.
static double post_ragged(Cascador* c, const RaggedJob& job, const RaggedChunk& ch, const RawDets<double>& dets,
                          const CppCall& call, jdaResultD* out, RowsOut<double>* rows = nullptr, int frame_offset = 0) {
  const double t0 = now_ms();
  const int L = c->hm.L, dim = c->hm.dim();
  const DevPlan& hp = job.pe->hp;
  std::vector<size_t> first(ch.n + 1, dets.gid.size());
  {
    size_t i = 0;
    for (int f = 0; f < ch.n; f++) {
      first[f] = i;
      while (i < dets.gid.size() && dets.gid[i] < ch.gid_base[f + 1]) i++;
    }
    first[ch.n] = i;
  }
  // candidates of image f in scan order -> Rect(x, y, win, win) (cascador.cpp:339)
  auto rects_of = [&](int f, std::vector<int>& rc) {
    const size_t a = first[f], cnt = first[f + 1] - a;
    rc.resize(cnt * 4);
    const int W = ch.widths[f], H = ch.heights[f];
    int l = 0;
    uint32_t lbase = ch.gid_base[f];
    int nx = 0, cntl = 0;
    auto level_grid = [&](int lv) {
      const DevLevel& d = hp.lv[lv];
      nx = (W - d.win) / d.step + 1;
      cntl = nx * ((H - d.win) / d.step + 1);
    };
    if (cnt) level_grid(0);
    for (size_t i = 0; i < cnt; i++) {          // gids ascend: levels are walked once
      const uint32_t g = dets.gid[a + i];
      while (g >= lbase + (uint32_t)cntl) { lbase += (uint32_t)cntl; l++; level_grid(l); }
      const uint32_t rel = g - lbase;
      const DevLevel& d = hp.lv[l];
      rc[4 * i] = (int)(rel % (uint32_t)nx) * d.step; rc[4 * i + 1] = (int)(rel / (uint32_t)nx) * d.step;
      rc[4 * i + 2] = d.win; rc[4 * i + 3] = d.win;
    }
  };
  if (rows) {
    // rows mode: NMS per image in parallel (the picks), then every image's rows written in place, in parallel
    std::vector<std::vector<int>> rcs((size_t)ch.n), picks((size_t)ch.n);
    parallel_for(ch.n, [&](int f) {
      const size_t a = first[f], cnt = first[f + 1] - a;
      rects_of(f, rcs[(size_t)f]);
      if (call.nms != 0) picks[(size_t)f] = nms_dialect_cpp(rcs[(size_t)f].data(), dets.score.data() + a, (int)cnt, call.overlap);   // cascador.cpp:444-446
      else { picks[(size_t)f].resize(cnt); std::iota(picks[(size_t)f].begin(), picks[(size_t)f].end(), 0); }                         // cascador.cpp:447-451
    }, dets.gid.size() < 6000);
    std::vector<size_t> row0((size_t)ch.n + 1, 0);
    for (int f = 0; f < ch.n; f++) row0[(size_t)f + 1] = row0[(size_t)f] + picks[(size_t)f].size();
    const size_t rw = (size_t)6 + dim;
    double* base = rows->grow(row0[(size_t)ch.n] * rw);
    parallel_for(ch.n, [&](int f) {
      const size_t a = first[f];
      const std::vector<int>& rc = rcs[(size_t)f];
      double* o = base + row0[(size_t)f] * rw;
      for (int k : picks[(size_t)f]) {
        o[0] = (double)(frame_offset + ch.i0 + f);
        for (int q = 0; q < 4; q++) o[1 + q] = (double)rc[4 * (size_t)k + q];
        o[5] = dets.score[a + k];
        std::memcpy(o + 6, dets.shape.data() + (a + k) * dim, dim * sizeof(double));
        relocate_dialect_cpp(o + 6, L, rc[4 * (size_t)k], rc[4 * (size_t)k + 1], rc[4 * (size_t)k + 2], rc[4 * (size_t)k + 3]);   // cascador.cpp:462-474
        o += rw;
      }
    }, row0[(size_t)ch.n] < 2000);
    return now_ms() - t0;
  }
  parallel_for(ch.n, [&](int f) {
    const size_t a = first[f], cnt = first[f + 1] - a;
    static thread_local std::vector<int> rc;
    rects_of(f, rc);
    emit_cpp_result(rc.data(), dets.score.data() + a, dets.shape.data() + a * dim, (int)cnt, L, call.overlap, call.nms != 0, &out[f]);
  }, dets.gid.size() < 6000);
  return now_ms() - t0;
}

static void add_stats(RunStats* a, const RunStats& b) {
  a->carts += b.carts; a->out += b.out; a->carts_scan += b.carts_scan; a->carts_scan_glb += b.carts_scan_glb;
  a->win_scan += b.win_scan; a->tail += b.tail; a->gpu_ms += b.gpu_ms; a->scan_ms += b.scan_ms;
  a->scan_launches += b.scan_launches; a->dense_passes += b.dense_passes; a->scan_fallbacks += b.scan_fallbacks; a->ws_regrows += b.ws_regrows;
  for (int t = 0; t < kMaxStages; t++) a->stage_done[t] += b.stage_done[t];
}

// What the two dialects do differently around a ragged job: the call's parameters, the per-image fallback, the
// post-processing and the result type.
struct RagSideC {
  using Real = float; using Result = jdaResult;
  float scale; int min_size, max_size; float th; const jdaDetectOptions* opt; jdaResult* out;
  RowsOut<float>* rows = nullptr; int frame_offset = 0;      // rows mode (post_ragged): `out` is the caller's scratch, left blank
  bool rows_mode() const { return rows != nullptr; }
  jdaStats* stats() const { return opt ? opt->stats : nullptr; }
  bool apply_th() const { return true; }
  Real final_th() const { return th; }
  void blank(int n, int L) const { for (int i = 0; i < n; i++) { out[i].n = 0; out[i].landmark_n = L; out[i].bboxes = nullptr; out[i].shapes = nullptr; out[i].scores = nullptr; } }
  bool filled(int i) const { return out[i].bboxes != nullptr; }
  void set_empty(int i, int L) const { out[i] = empty_result(L); }
  void describe(RaggedJob* job) const { job->cpp = false; job->scale = scale; job->min_size = min_size; job->max_size = max_size; }
  bool usable(Cascador*) const { return true; }
  // one image as a pass of its own (detect.cpp)
  int one_image(Cascador* c, const unsigned char* host, const uint8_t* dev, int W, int H, jdaStats* st1, int i) const {
    jdaDetectOptions o1;
    if (opt) o1 = *opt; else jdaDetectOptionsInit(&o1);
    o1.stats = st1; o1.hip_stream = nullptr;
    const unsigned char* one[1] = {host};
    return host ? detect_c_device(c, nullptr, 0, 1, W, H, scale, min_size, max_size, th, &o1, out + i, one)
                : detect_c_device(c, dev, (size_t)W * H, 1, W, H, scale, min_size, max_size, th, &o1, out + i);
  }
  double post(Cascador* c, const RaggedJob& job, const RaggedChunk& ch, const RawDets<float>& dets) const { return post_ragged(c, job, ch, dets, opt, out + ch.i0, rows, frame_offset); }
  bool device_post(Cascador* c, int n_imgs) const { return c->kn.device_post >= 1 && n_imgs >= c->kn.device_post_min_frames; }
  bool post_nms() const { return !opt || opt->nms; }
  float post_overlap() const { return opt ? opt->nms_overlap : 0.3f; }
};
struct RagSideCpp {
  using Real = double; using Result = jdaResultD;
  CppCall call; jdaStats* st; jdaResultD* out;
  jdaStats* stats() const { return st; }
  bool apply_th() const { return false; }                  // Validate has no final threshold (cascador.cpp:166-211)
  Real final_th() const { return 0.0; }
  void blank(int n, int L) const { for (int i = 0; i < n; i++) { out[i].n = 0; out[i].landmark_n = L; out[i].rects = nullptr; out[i].shapes = nullptr; out[i].scores = nullptr; } }
  bool filled(int i) const { return out[i].rects != nullptr; }
  void set_empty(int i, int L) const { out[i] = empty_result_d(L); }
  void describe(RaggedJob* job) const { job->cpp = true; job->call = call; }
  RowsOut<double>* rows = nullptr; int frame_offset = 0;     // rows mode (post_ragged): `out` is the caller's scratch, left blank
  bool rows_mode() const { return rows != nullptr; }
  bool usable(Cascador* c) const { return cpp_model_complete(c); }
  int one_image(Cascador* c, const unsigned char* host, const uint8_t* dev, int W, int H, jdaStats* st1, int i) const {
    const unsigned char* one[1] = {host};
    return host ? detect_cpp_device(c, nullptr, 0, 1, W, H, call, st1, out + i, one)
                : detect_cpp_device(c, dev, (size_t)W * H, 1, W, H, call, st1, out + i);
  }
  double post(Cascador* c, const RaggedJob& job, const RaggedChunk& ch, const RawDets<double>& dets) const { return post_ragged(c, job, ch, dets, call, out + ch.i0, rows, frame_offset); }
  bool device_post(Cascador*, int) const { return false; }  // (k_post is dialect C's NMS; the multimap NMS runs on the host)
  bool post_nms() const { return call.nms != 0; }
  float post_overlap() const { return (float)call.overlap; }
};

// A ragged job: images of different sizes, in host memory (host_imgs) or on the device (d_base + d_offsets).
template <typename Side>
static int detect_ragged_t(Cascador* c, const unsigned char* const* host_imgs, const uint8_t* d_base, const size_t* d_offsets,
                           const int* widths, const int* heights, int n, const Side& side) {
  using Real = typename Side::Real;
  const double t_call = now_ms();
  if (!c || !side.out || n < 0 || !widths || !heights || (!host_imgs && !(d_base && d_offsets))) { fail("bad arguments"); return -1; }
  const int L = c->hm.L;
  side.blank(n, L);
  if (!side.usable(c)) return -1;
  if (n == 0) return 0;
  {
    std::lock_guard<std::mutex> lk(c->mu);
    if (!ensure_device(c) || !upload_model<Real>(c)) return -1;
  }
  if (c->kn.debug_times == 1) fprintf(stderr, "[jda] ragged job: device and model ready at %.3f ms\n", now_ms() - t_call);
  double tm[6] = {now_ms() - t_call, 0, 0, 0, 0, 0};     // debug_times = 2: one line at the end (prints cost tens of microseconds each)
  RunStats total;
  long long patch_n = 0;
  double post_ms = 0;
  jdaStats st1;
  auto finish = [&]() {
    fill_stats(side.stats(), total, patch_n, c->hm.T, c->hm.K, post_ms);
    if (side.stats()) side.stats()->call_ms = now_ms() - t_call;
    return 0;
  };
  // per-image passes: models the ragged scan does not cover (multi-scale split nodes, levels without a tile), and
  // cascades that reject so little that the dense kernel is the right tool
  auto fallback = [&]() -> int {
    for (int i = 0; i < n; i++) {
      const int W = widths[i], H = heights[i];
      if (W <= 0 || H <= 0) { fail("image " + std::to_string(i) + " has no pixels"); return -1; }
      if (host_imgs && !host_imgs[i]) { fail("null image pointer"); return -1; }
      const int rc = side.one_image(c, host_imgs ? host_imgs[i] : nullptr, host_imgs ? nullptr : d_base + d_offsets[i], W, H, &st1, i);
      if (rc != 0) return -1;
      total.carts += st1.cart_total_n; total.out += st1.face_patch_n; total.carts_scan += st1.scan_cart_n;
      total.win_scan += st1.scan_patch_n; total.tail += st1.handoff_n; total.gpu_ms += st1.gpu_ms; total.scan_ms += st1.scan_ms;
      total.scan_launches += st1.scan_launches; total.dense_passes += st1.dense_passes; total.scan_fallbacks += st1.scan_fallbacks; total.ws_regrows += st1.ws_regrows;
      for (int t = 0; t < 16 && t < kMaxStages; t++) total.stage_done[t] += st1.stage_done_n[t];
      patch_n += st1.patch_n; post_ms += st1.host_ms;
    }
    return finish();
  };
  if (c->hm.multi_scale()) return fallback();
  RaggedJob job;
  job.n = n; job.widths = widths; job.heights = heights; job.host_imgs = host_imgs; job.d_base = d_base; job.d_offsets = d_offsets;
  side.describe(&job);
  const int prep = ragged_prepare(c, &job);
  PlanPin pin{c, job.pe};
  tm[1] = now_ms() - t_call;
  if (c->kn.debug_times == 1) fprintf(stderr, "[jda] ragged job: prepared at %.3f ms\n", now_ms() - t_call);
  if (prep < 0) return -1;
  if (prep > 0) return fallback();
  if (job.levels.levels.empty()) {                                // no image holds a window: n empty results
    for (int i = 0; i < n && !side.rows_mode(); i++) side.set_empty(i, L);
    return finish();
  }

  // ---- chunks: as many images as make ragged_chunk_windows windows (<= 65535 images, the queues pack the index
  //      in 16 bits), walked through up to three lanes as a software pipeline: while the GPU works on chunks i-1 and i-2
  //      the host builds and issues chunk i and post-processes chunk i-3 ----
  const DevPlan& hp = job.pe->hp;
  // (dialect CPP scans ~3.4x the windows of dialect C on the same images -- 20-pixel windows, 5-pixel step -- in levels of
  // few workgroups each: larger chunks, measured 42.3 -> 38.1 ms for the FDDB-shaped job)
  const long long chunk_windows = job.cpp ? c->kn.ragged_chunk_windows_cpp : c->kn.ragged_chunk_windows;
  // host images: uploaded by a helper thread, below (packed = every image right behind the one before in memory)
  bool helper = host_imgs != nullptr && c->kn.ragged_uploader != 0, packed = helper;
  std::vector<size_t> tight(helper ? (size_t)n + 1 : 0, 0);       // image i at tight[i] of the job's tight image buffer
  for (int i = 0; i < n && helper; i++) {
    if (!host_imgs[i] || widths[i] <= 0 || heights[i] <= 0) { helper = false; break; }
    if (i > 0 && host_imgs[i] != host_imgs[i - 1] + (size_t)widths[i - 1] * heights[i - 1]) packed = false;
    tight[i + 1] = tight[i] + (size_t)widths[i] * heights[i];
  }
  std::vector<int> starts;
  {
    long long wsum = 0; int cnt = 0;
    const std::vector<long long>& wins = job.wins;
    long long total = 0;
    for (int i = 0; i < n; i++) {
      if (wins[(size_t)i] > 0x7fffffffLL) { fail("image has too many windows"); return -1; }
      total += wins[(size_t)i];
    }
    // A chunk is at most ragged_chunk_windows; a job of fewer than three such chunks is cut into three (down to
    // ragged_chunk_min_windows each), so that its lanes overlap too: the scan of one chunk next to the latency-bound
    // finishing kernels of another (a 356-image shard of the FDDB-sized job as one chunk: 1.58 ms, as three: 1.50)
    const long long split = std::max<long long>(1, c->kn.ragged_split);
    const long long full = std::max<long long>(1, std::min<long long>(chunk_windows,
                                                                      std::max<long long>(c->kn.ragged_chunk_min_windows, (total + split - 1) / split)));
    long long target = full;
    const bool single = !helper && total <= c->kn.ragged_single_windows && n <= 65535;     // (r06: see Pass::issue_scan_ragged)
    starts.push_back(0);
    for (int i = 0; i < n; i++) {
      const long long wi = wins[(size_t)i];
      // (a job whose pixels still have to come over the link starts with a quarter and a half chunk: the GPU has work
      // after a quarter of a chunk's upload time instead of a whole one)
      if (helper) target = starts.size() == 1 ? full / 4 : (starts.size() == 2 ? full / 2 : full);
      if (cnt > 0 && !single && (wsum + wi > target || cnt >= 65535)) { starts.push_back(i); wsum = 0; cnt = 0; }
      wsum += wi; cnt++;
    }
    starts.push_back(n);
  }
  const int n_chunks = (int)starts.size() - 1;
  const int lanes = std::min(std::min((int)std::max<long long>(1, std::min<long long>(8, c->kn.ragged_lanes)), n_chunks), (int)std::max<long long>(1, c->kn.max_lanes));
  struct Slot { bool busy = false; RaggedChunk ch; Pass<Real> pass; RawDets<Real> dets; RunStats rs; };
  std::vector<Slot> slots(lanes);
  LaneSet held(c);
  if (!held.take(lanes, n_chunks > 1 ? (size_t)chunk_windows : 0, true)) return -1;
  bool ok = true;

  // ---- host images, several chunks: a helper thread brings chunk after chunk into a buffer of the job (the first
  //      lane's) on the cascador's upload stream and waits for each upload on the host; this thread builds tables,
  //      enqueues passes and post-processes meanwhile, and only waits for a chunk's pixels right before it enqueues that
  //      chunk.  (Issued here, every pageable upload blocked this thread for a millisecond, and staging 2,845 separate
  //      arrays with one memcpy loop took longer than the GPU needs for the job.)  Images that lie back to back go up
  //      straight from the caller's memory; separate arrays are gathered into two pinned buffers (the first two lanes')
  //      by `ragged_stage_threads` copy threads, chunk k+1 while chunk k is on the link. ----
  struct Uploader {
    std::thread th;
    std::mutex mu; std::condition_variable cv;
    int ready = 0;                 // chunks [0, ready) are on the device
    bool failed = false, stop = false;
    std::string err;
    Cascador* c = nullptr;
    // Leaving the job (also on its error paths, before the lanes go back to the pool and the caller frees its images):
    // the helper is stopped and joined, and whatever it had queued on the cascador's upload stream -- a copy out of the
    // caller's memory or a lane's pinned buffer into a lane's device buffer -- is waited for.
    ~Uploader() {
      { std::lock_guard<std::mutex> lk(mu); stop = true; }
      if (!th.joinable()) return;
      th.join();
      std::lock_guard<std::mutex> lk(c->h2d_mu);
      if (c->h2d) (void)hipStreamSynchronize(c->h2d);
    }
  } up;
  up.c = c;
  if (helper && n_chunks > 1 && lanes > 1 && held.v[0]->rag_raw.reserve(tight[n] + 16)) {
    size_t most = 0;
    job.raw_off.assign(n_chunks + 1, 0);
    for (int k = 0; k < n_chunks; k++) { job.raw_off[k + 1] = tight[starts[k + 1]]; most = std::max(most, job.raw_off[k + 1] - job.raw_off[k]); }
    if (packed || (held.v[0]->h_raw.reserve(most + 16) && held.v[1]->h_raw.reserve(most + 16))) {
      job.d_job_raw = (uint8_t*)held.v[0]->rag_raw.p;
      const int dev = c->device;
      const int copy_threads = (int)std::max<long long>(1, std::min<long long>(16, c->kn.ragged_stage_threads));
      uint8_t* stage[2] = {(uint8_t*)held.v[0]->h_raw.p, (uint8_t*)held.v[1]->h_raw.p};
      auto uploader_body = [&, dev, copy_threads, stage]() {
        bool good = hipSetDevice(dev) == hipSuccess;
        auto publish = [&](int k_done) {
          std::lock_guard<std::mutex> lk(up.mu);
          if (good) up.ready = k_done;
          else { up.failed = true; up.err = std::string("upload of a ragged chunk failed: ") + hipGetErrorString(hipGetLastError()); }
          up.cv.notify_all();
        };
        auto drain = [&]() {          // the uploads queued so far are on the device
          std::lock_guard<std::mutex> lk(c->h2d_mu);
          good = good && hipStreamSynchronize(c->h2d) == hipSuccess;
        };
        for (int k = 0; k < n_chunks && good; k++) {
          { std::lock_guard<std::mutex> lk(up.mu); if (up.stop) return; }
          const double t_up = now_ms();
          const size_t bytes = job.raw_off[k + 1] - job.raw_off[k];
          const uint8_t* src = host_imgs[0] + job.raw_off[k];
          if (!packed) {
            // gather the chunk's images into pinned buffer k & 1 (its last upload, chunk k-2, was drained one round ago)
            uint8_t* dst = stage[k & 1];
            const int a = starts[k], b = starts[k + 1];
            auto copy_range = [&](int i0, int i1) {
              for (int i = i0; i < i1; i++) std::memcpy(dst + (tight[i] - tight[a]), host_imgs[i], tight[i + 1] - tight[i]);
            };
            std::vector<std::thread> ts;
            int i0 = a;
            for (int t = 0; t < copy_threads && i0 < b; t++) {
              const size_t upto = tight[a] + bytes * (size_t)(t + 1) / (size_t)copy_threads;
              int i1 = i0;
              while (i1 < b && (tight[i1 + 1] <= upto || t == copy_threads - 1)) i1++;
              if (i1 == i0) continue;
              if (t == copy_threads - 1 || i1 == b) { copy_range(i0, b); i0 = b; }
              else {
                try { ts.emplace_back(copy_range, i0, i1); } catch (...) { copy_range(i0, i1); }   // (no thread to be had: copy here)
                i0 = i1;
              }
            }
            if (i0 < b) copy_range(i0, b);
            for (auto& t : ts) t.join();
            src = dst;
          }
          if (k > 0 && !packed) { drain(); publish(k); }           // chunk k-1 has arrived while this one was gathered
          {
            std::lock_guard<std::mutex> lk(c->h2d_mu);
            if (!c->h2d) good = (c->h2d = c->streams.take(StreamPool::kSide, StreamPool::kNone, nullptr)) != nullptr;
            good = good && hipMemcpyAsync(job.d_job_raw + job.raw_off[k], src, bytes, hipMemcpyHostToDevice, c->h2d) == hipSuccess;
          }
          if (packed || k == n_chunks - 1) { drain(); publish(k + 1); }
          if (c->kn.debug_times) fprintf(stderr, "[jda] ragged upload %d: %.3f MB at %.3f..%.3f ms\n", k, bytes / 1e6, t_up - t_call, now_ms() - t_call);
        }
        if (!good) publish(0);
      };
      // (a thread body: an exception that left it would end the process -- a failed allocation in there fails the job)
      auto uploader = [&up, uploader_body]() {      // (the body by VALUE: it is a local of this block, the thread outlives the block)
        try { uploader_body(); }
        catch (...) {
          std::lock_guard<std::mutex> lk(up.mu);
          up.failed = true;
          try { up.err = "upload of a ragged chunk failed: C++ exception (out of host memory?)"; } catch (...) {}
          up.cv.notify_all();
        }
      };
      try { up.th = std::thread(uploader); }
      catch (...) { job.d_job_raw = nullptr; }          // (no thread to be had: the chunks upload themselves, as without a helper)
    }
  }
  auto collect = [&](Slot& sl) -> bool {
    Pass<Real>& p = sl.pass;
    sl.busy = false;
    if (!p.after_tail() || !p.issue_counters() || !p.after_counters() || !p.collect()) return false;
    float ms_scan = 0, ms_all = 0;
    if (p.timed) {
      (void)hipEventElapsedTime(&ms_scan, p.ev[1], p.ev[2]);
      (void)hipEventElapsedTime(&ms_all, p.ev[0], p.ev[3]);
    }
    sl.rs.scan_ms += ms_scan; sl.rs.gpu_ms += ms_all;
    post_ms += side.post(c, job, sl.ch, sl.dets);
    add_stats(&total, sl.rs);
    patch_n += sl.ch.windows;
    return true;
  };
  for (int ci = 0; ci < n_chunks && ok; ci++) {
    const int lane = ci % lanes;
    Slot& sl = slots[lane];
    if (sl.busy && !collect(sl)) { ok = false; break; }
    sl.dets = RawDets<Real>(); sl.rs = RunStats(); sl.rs.timed = side.stats() != nullptr;
    Lane* ln = held.v[lane];
    const double t_b = now_ms();
    if (!ragged_build_images(job, starts[ci], starts[ci + 1] - starts[ci], ln, &sl.ch)) { ok = false; break; }
    if (sl.ch.windows > 0 && !job.host_imgs) {
      // images resident on the device: their records go up and k_repack starts NOW, the segments and the block map --
      // most of the table-building time -- are built while it runs (r06: 35 us off a 356-image job's critical path)
      const bool timed = side.stats() != nullptr || c->kn.debug_times;
      const RaggedChunk& ch = sl.ch;
      uint8_t* tab = (uint8_t*)ln->rag_tab.p;
      if ((timed && hipEventRecord(ln->ev[0], ln->stream) != hipSuccess) ||
          hipMemcpyAsync(tab, ln->h_tab.p, ch.images_bytes, hipMemcpyHostToDevice, ln->stream) != hipSuccess ||
          launch_repack(ch.d_raw, (uint8_t*)ln->rag_frames.p, (const RagImg*)(tab + ch.off_rimg), ch.n, ch.max_h, ch.pitch, ln->stream) != hipSuccess) {
        fail(std::string("queueing the images of a ragged chunk failed: ") + hipGetErrorString(hipGetLastError())); ok = false; break;
      }
      sl.ch.images_issued = true;
    }
    if (!ragged_build_chunk(c, job, starts[ci], starts[ci + 1] - starts[ci], ln, &sl.ch)) { ok = false; break; }
    tm[2] = now_ms() - t_call;
    if (c->kn.debug_times == 1) fprintf(stderr, "[jda] ragged chunk %d: tables built %.3f..%.3f ms (%d blocks, %d segments)\n", ci, t_b - t_call, now_ms() - t_call, sl.ch.n_blk, sl.ch.n_segs);
    if (sl.ch.windows == 0) {                        // images too small for any window
      post_ms += side.post(c, job, sl.ch, sl.dets);
      continue;
    }
    // workspace: every lane holds a whole chunk (its previous chunk has been collected above)
    {
      const size_t want = n_chunks > 1 ? std::max<size_t>((size_t)sl.ch.windows, (size_t)std::min<long long>(chunk_windows, 0x7fffffffLL))
                                       : (size_t)sl.ch.windows;
      bool want_dense = false;
      const QueueCaps qc = plan_queue_caps(c, job.pe, want, false, &want_dense);
      if (!ensure_workspace<Real>(ln, want, false, c->hm.dim(), qc.q, qc.m, false)) { ok = false; break; }      // (a ragged pass never runs dense)
    }
    Pass<Real>& p = sl.pass;
    p = Pass<Real>();
    p.c = c; p.pe = job.pe; p.trace = nullptr; p.dets = &sl.dets; p.rs = &sl.rs; p.apply_th = side.apply_th(); p.th = side.final_th(); p.multi = false;
    p.solo = lanes == 1;
    p.bind(ln, lane, nullptr);
    p.f0 = 0; p.nf = sl.ch.n; p.rag = &sl.ch;
    if (side.device_post(c, sl.ch.n)) {
      p.want_post = true; p.post_nms = side.post_nms(); p.post_overlap = side.post_overlap();
      sl.dets.p_n.assign((size_t)sl.ch.n, -1); sl.dets.p_first.assign((size_t)sl.ch.n, 0);
    }
    p.w.half = nullptr; p.w.quarter = nullptr; p.w.half_stride = p.w.quarter_stride = 0;
    p.w.hw = p.w.hh = p.w.qw = p.w.qh = 0;
    if (job.d_job_raw) {
      const double t_w = now_ms();
      std::unique_lock<std::mutex> lk(up.mu);
      up.cv.wait(lk, [&] { return up.ready > ci || up.failed; });
      if (c->kn.debug_times) fprintf(stderr, "[jda] ragged chunk %d: waited for its pixels %.3f..%.3f ms\n", ci, t_w - t_call, now_ms() - t_call);
      if (up.failed) { fail(up.err); ok = false; break; }
      sl.ch.d_uploaded = job.d_job_raw + job.raw_off[ci];
    }
    if (!p.issue_scan(nullptr, 0, nullptr, 0, nullptr)) { ok = false; break; }
    sl.busy = true;
  }
  tm[3] = now_ms() - t_call;
  if (c->kn.debug_times == 1) fprintf(stderr, "[jda] ragged job: all chunks issued at %.3f ms\n", now_ms() - t_call);
  // drain in chunk order
  for (int k = 0; k < lanes && ok; k++) {
    Slot& sl = slots[(n_chunks + k) % lanes];
    if (sl.busy && !collect(sl)) ok = false;
  }
  if (!ok) {
    for (Lane* l : held.v) (void)hipStreamSynchronize(l->stream);
    return -1;
  }
  for (int i = 0; i < n && !side.rows_mode(); i++)
    if (!side.filled(i)) side.set_empty(i, L);     // (chunks fill every image; belt and braces)
  if (c->kn.debug_times == 1) fprintf(stderr, "[jda] ragged job: done at %.3f ms (post-processing %.3f ms of it)\n", now_ms() - t_call, post_ms);
  if (c->kn.debug_times >= 2) fprintf(stderr, "[jda] ragged job: ready %.3f prepared %.3f last tables %.3f issued %.3f done %.3f ms (post %.3f; prepare: levels %.3f grids %.3f plan %.3f)\n", tm[0], tm[1], tm[2], tm[3], now_ms() - t_call, post_ms,
                                       t_prep_marks[0] - t_call, t_prep_marks[1] - t_call, t_prep_marks[2] - t_call);
  return finish();
}

int detect_ragged(Cascador* c, const unsigned char* const* host_imgs, const uint8_t* d_base, const size_t* d_offsets,
                  const int* widths, const int* heights, int n, float scale, int min_size, int max_size, float th,
                  const jdaDetectOptions* opt, jdaResult* out) {
  return detect_ragged_t(c, host_imgs, d_base, d_offsets, widths, heights, n, RagSideC{scale, min_size, max_size, th, opt, out});
}

// The same job with the results as one matrix of rows (jdaDetectBatchRagged[Device]Rows).  Jobs that run image by image
// inside (multi-scale models, dense cascades) fill the scratch results; they are packed here.
int detect_ragged_rows(Cascador* c, const unsigned char* const* host_imgs, const uint8_t* d_base, const size_t* d_offsets,
                       const int* widths, const int* heights, int n, float scale, int min_size, int max_size, float th,
                       const jdaDetectOptions* opt, int frame_offset, RowsOut<float>* rows) {
  std::vector<jdaResult> scratch((size_t)std::max(n, 1));
  RagSideC side{scale, min_size, max_size, th, opt, scratch.data()};
  side.rows = rows; side.frame_offset = frame_offset;
  rows->n = 0;
  const int rc = detect_ragged_t(c, host_imgs, d_base, d_offsets, widths, heights, n, side);
  bool any = false;
  for (int i = 0; i < n; i++) any = any || scratch[(size_t)i].bboxes != nullptr;
  if (any && rc == 0) {                              // (the per-image fallback ran: nothing has been appended yet)
    rows->n = 0;
    const int dim = c->hm.dim();
    for (int i = 0; i < n; i++) {
      const jdaResult& r = scratch[(size_t)i];
      for (int j = 0; j < r.n && r.bboxes; j++) {
        float* o = rows->grow(5 + (size_t)dim);
        o[0] = (float)(frame_offset + i); o[1] = (float)r.bboxes[3 * j]; o[2] = (float)r.bboxes[3 * j + 1]; o[3] = (float)r.bboxes[3 * j + 2];
        o[4] = r.scores[j];
        std::memcpy(o + 5, r.shapes + (size_t)j * dim, dim * sizeof(float));
      }
    }
  }
  for (int i = 0; i < n; i++) { std::free(scratch[(size_t)i].bboxes); std::free(scratch[(size_t)i].shapes); std::free(scratch[(size_t)i].scores); }
  return rc;
}

// Dialect CPP (jdaDetectBatchCppRagged[Device]Rows): rows [frame, x, y, w, h, score, shape] of doubles -- a job of 2,845
// FDDB-sized images keeps 32 k candidates, 15 MB of rows: through n jdaResultDs and jdaResultsDPack they were copied
// three times and cost the binding 4.3 ms of a 38-ms job.
int detect_ragged_cpp_rows(Cascador* c, const unsigned char* const* host_imgs, const uint8_t* d_base, const size_t* d_offsets,
                           const int* widths, const int* heights, int n, const CppCall& call, jdaStats* stats, int frame_offset,
                           RowsOut<double>* rows) {
  std::vector<jdaResultD> scratch((size_t)std::max(n, 1));
  RagSideCpp side{call, stats, scratch.data()};
  side.rows = rows; side.frame_offset = frame_offset;
  rows->n = 0;
  const int rc = detect_ragged_t(c, host_imgs, d_base, d_offsets, widths, heights, n, side);
  bool any = false;
  for (int i = 0; i < n; i++) any = any || scratch[(size_t)i].rects != nullptr;
  if (any && rc == 0) {                              // (the per-image fallback ran)
    rows->n = 0;
    const int dim = c->hm.dim();
    for (int i = 0; i < n; i++) {
      const jdaResultD& r = scratch[(size_t)i];
      for (int j = 0; j < r.n && r.rects; j++) {
        double* o = rows->grow(6 + (size_t)dim);
        o[0] = (double)(frame_offset + i);
        for (int k = 0; k < 4; k++) o[1 + k] = (double)r.rects[4 * j + k];
        o[5] = r.scores[j];
        std::memcpy(o + 6, r.shapes + (size_t)j * dim, dim * sizeof(double));
      }
    }
  }
  for (int i = 0; i < n; i++) { std::free(scratch[(size_t)i].rects); std::free(scratch[(size_t)i].shapes); std::free(scratch[(size_t)i].scores); }
  return rc;
}

int detect_ragged_cpp(Cascador* c, const unsigned char* const* host_imgs, const uint8_t* d_base, const size_t* d_offsets,
                      const int* widths, const int* heights, int n, const CppCall& call, jdaStats* stats, jdaResultD* out) {
  return detect_ragged_t(c, host_imgs, d_base, d_offsets, widths, heights, n, RagSideCpp{call, stats, out});
}

}  // namespace jda
