// libjda.so, host side: how k_scan covers each pyramid level (tile chooser) and the cache of scan plans.
#include "host.h"

namespace jda {

// ---------------------------------------------------------------- tiling of levels

// Chooses, per level, how k_scan covers it (DESIGN.md "LDS tiles"): the tile of windows (tw x th, at most
// 512) that share one LDS pixel tile, or the global-pixel mode for windows that do not fit LDS.
//
// Candidates are every tile shape whose workgroup fits LDS; each is priced with a small throughput model
// (CU clocks per frame, constants from the r02 kernel traces) and the cheapest wins:
//   per workgroup   c_fix + pix_bytes / c_bw + slots * c_win        (table load + barriers, tile load, cart walks)
//   per CU          divided by min(1, (waves per CU / w_sat)^alpha)  (latency hiding needs resident waves)
// slots = lanes the tile occupies in phase 0: windows rounded up to whole waves, or to the power of two the
// pair phases pad to for tiles of few windows.  Tile origins need not be multiples of 16 pixels: the
// LDS-DMA loader starts at the 16-byte chunk below and the pitch covers the lead-in.
// JDA_TILES="win:twxth,win:twxth" forces shapes (experiments); JDA_DEBUG_TILES=1 prints the choice.
struct TileChoice { int mode = 0, tw = 1, th = 1, pitch = 0, pix = 0, lds = 0, block = 256; double cost = 0; };

static TileChoice choose_tile(const Level& s, int width, const HostModel& hm, int real_bytes, int chunk, int cp_max,
                              bool ragged = false) {
  static const double c_win = (double)env_ll("JDA_TILE_CWIN", 23), c_bw = (double)env_ll("JDA_TILE_CBW", 32),
                      c_fix = (double)env_ll("JDA_TILE_CFIX", 1500), w_sat = (double)env_ll("JDA_TILE_WSAT", 20),
                      alpha = (double)env_ll("JDA_TILE_ALPHA_PCT", 70) / 100.0;
  const int lds_cu = 160 * 1024;
  static const int lds_max = (int)std::min<long long>(lds_cu, env_ll("JDA_SCAN_LDS_MAX", lds_cu));
  static const char* const tiles_env = std::getenv("JDA_TILES");      // (experiments; read once per process)
  int force_tw = 0, force_th = 0;
  if (const char* e = tiles_env) {
    for (const char* p = e; p && *p;) {
      int w = 0, a = 0, b = 0;
      if (std::sscanf(p, "%d:%dx%d", &w, &a, &b) == 3 && w == s.win) { force_tw = a; force_th = b; }
      p = std::strchr(p, ',');
      if (p) p++;
    }
  }
  TileChoice best;
  const int fixed256 = (int)scan_lds_bytes(0, chunk, hm.node_n(), hm.leaf_n(), real_bytes, true, 256);
  const int fixed512 = (int)scan_lds_bytes(0, chunk, hm.node_n(), hm.leaf_n(), real_bytes, true, 512);
  const int tw_hi = std::min(s.nx, 128), th_hi = std::min(s.ny, 128);
  for (int th = 1; th <= th_hi; th++) {
    for (int tw = 1; tw <= tw_hi; tw++) {
      if (force_tw && (tw != force_tw || th != force_th)) continue;
      const int n_tile = tw * th;
      if (n_tile > 512) break;
      // no point in tiles smaller than a pair-phase round unless the level itself is that small
      if (!force_tw && n_tile < 16 && n_tile < s.nx * s.ny && (long long)s.win * s.win < 64 * 1024) continue;
      const int tiles_x = (s.nx + tw - 1) / tw, tiles_y = (s.ny + th - 1) / th;
      const int pw = s.win + (tw - 1) * s.step, ph = s.win + (th - 1) * s.step;
      int xs = 0;
      if (ragged) {
        // images of any width share the tile's LDS pitch: the worst lead-in of a tile origin x0 = tx * tw * step
        // (and an image may re-cut the tile narrower, ragged_tile: any lead-in below 16 can occur)
        xs = 15;
      } else {
        for (int tx = 0; tx < tiles_x; tx++) xs = std::max(xs, (tx * tw * s.step) & 15);
        static const long long force_xs = env_ll("JDA_TILE_XS", -1);     // (experiment: the ragged chooser's worst-case lead-in)
        if (force_xs >= 0) xs = (int)force_xs;
      }
      int pitch = (xs + pw + 15) & ~15;
      if ((pitch & 127) == 0) pitch += 16;          // keep tile rows off a 32-bank multiple
      const long long pix = (long long)pitch * ph;
      const int block = n_tile > 256 ? 512 : 256;
      const long long lds = (block == 512 ? fixed512 : fixed256) + ((pix + 15) & ~15LL);
      if (lds > lds_max) continue;                   // (the pitch is not monotonic in tw: a wider tile can fit again)
      const long long max_off = (long long)(s.win - 1) * pitch + s.win - 1 + 15;
      if (max_off >= (1LL << kS0GlobalOffBits)) continue;
      const int mode = max_off <= 65535 ? 1 : 3;
      // (workgroups per CU in LDS granules, host.h.  r06 A/B against lds_cu / lds, also with the LDS of the untraced launch:
      // headline, configs[2] and the FDDB-shaped job equal within noise -- the shapes chosen sit away from the boundaries)
      const int wgs = (int)std::min<long long>(lds_wgs_per_cu(lds), 32 / (block / 64));
      const double waves = (double)wgs * (block / 64);
      int slots = (n_tile + 63) & ~63;
      if (n_tile <= cp_max) { slots = 16; while (slots < n_tile) slots *= 2; }
      const double eff = std::min(1.0, std::pow(waves / w_sat, alpha));
      const double cost = (double)tiles_x * tiles_y * (c_fix + (double)pix / c_bw + (double)slots * c_win) / eff;
      if (best.mode == 0 || cost < best.cost) {
        best.mode = mode; best.tw = tw; best.th = th; best.pitch = pitch; best.pix = (int)pix; best.lds = (int)lds;
        best.block = block; best.cost = cost;
      }
    }
  }
  (void)width;
  return best;
}

// ragged: sp holds the global level list of a ragged batch with NOMINAL grids (the mean nx, ny over the images that
// have the level) and the common row pitch as its width; the shapes must suit every image
void assign_tiles(const ScanPlan& sp, const HostModel& hm, const Knobs& kn, bool fast_scan, int real_bytes, PlanEntry* pe,
                         bool ragged) {
  DevPlan& hp = pe->hp;
  hp.n_levels = (int)sp.levels.size();
  hp.width = sp.width; hp.height = sp.height; hp.windows = (int)sp.windows;
  int table = 0;
  pe->any_untiled = false;
  const int handoff = (int)kn.handoff;
  const int chunk = std::min(std::min(hm.K, handoff), scan_handoff_cap(hm.node_n(), hm.leaf_n(), real_bytes));
  const int cp_max = (int)std::max<long long>(0, std::min<long long>(256, kn.cp_max));
  // a level's cost per window in global-pixel mode, in the units of choose_tile (r01: 0.38 ms for 952 k windows)
  const double glb_per_window = (double)kn.tile_cglb;
  for (int i = 0; i < hp.n_levels; i++) {
    const Level& s = sp.levels[i];
    DevLevel& d = hp.lv[i];
    d.win = s.win; d.step = s.step; d.nx = s.nx; d.ny = s.ny; d.base = (int)s.base;
    d.tiled = 0; d.tw = d.th = 1; d.tiles_x = d.tiles_y = 0; d.pitch = 0; d.s0_table = 0;
    const bool glb_ok = kn.no_global_scan == 0 &&
                        (long long)(s.win - 1) * sp.width + s.win - 1 < (1LL << kS0GlobalOffBits);
    if (fast_scan) {
      const TileChoice t = (kn.no_lds_scan || s.win > kn.lds_win_max) ? TileChoice() : choose_tile(s, sp.width, hm, real_bytes, chunk, cp_max, ragged);
      if (t.mode && (!glb_ok || t.cost <= glb_per_window * (double)s.nx * s.ny)) {
        d.tiled = t.mode; d.tw = t.tw; d.th = t.th; d.pitch = t.pitch;
      } else if (glb_ok) {
        // no LDS tile: k_scan reads the frame through L1/L2 (the offsets fit the packed node).  The tile is only a
        // grouping of up to 512 windows per workgroup here: the shape that wastes the fewest lane slots of the
        // first phase (a fixed 32 x 16 filled about half of them on the big-window levels of 640x480)
        d.tiled = 2; d.tw = 32; d.th = 16; d.pitch = sp.width;
        if (kn.glb_tile_fit) {
          long long best = -1;
          for (int th = 1; th <= std::min(s.ny, 512); th++)
            for (int tw = 1; tw <= std::min(s.nx, 512) && tw * th <= 512; tw++) {
              const int n_tile = tw * th;
              int slots = (n_tile + 63) & ~63;
              if (n_tile <= cp_max) { slots = 16; while (slots < n_tile) slots *= 2; }
              const long long tiles = (long long)((s.nx + tw - 1) / tw) * ((s.ny + th - 1) / th);
              const long long cost = tiles * (slots + 96);          // (+ a fixed cost per workgroup: table load, barriers)
              if (best < 0 || cost < best) { best = cost; d.tw = tw; d.th = th; }
            }
        }
      }
      if (kn.debug_tiles)
        std::fprintf(stderr, "[jda] level %d win %d step %d windows %dx%d: mode %d tile %dx%d pitch %d pix %d lds %d block %d cost/window %.0f\n",
                     i, s.win, s.step, s.nx, s.ny, d.tiled, d.tw, d.th, d.pitch, t.pix, t.lds, t.block,
                     t.mode ? t.cost / ((double)s.nx * s.ny) : 0.0);
    }
    if (!d.tiled) { pe->any_untiled = true; continue; }
    d.tiles_x = (s.nx + d.tw - 1) / d.tw;
    d.tiles_y = (s.ny + d.th - 1) / d.th;
    d.s0_table = table;
    table += hm.K * hm.node_n();
  }
}

// The plan of (frame size, call parameters), built on first use (host.h).
bool get_plan(Cascador* c, std::unique_lock<std::mutex>& lk, const PlanKey& key, const ScanPlan& sp, int dialect, PlanEntry** out,
              bool ragged) {
  auto it = c->plans.find(key);
  if (it != c->plans.end()) {
    PlanEntry& e = it->second;          // (map nodes do not move, and a pinned entry is not evicted)
    e.last_use = ++c->plan_clock; e.pins++;
    while (e.building) c->plan_cv.wait(lk);
    if (e.failed) {
      if (--e.pins == 0) c->plans.erase(key);
      fail("the scan plan of this frame size could not be built (see the first caller's error)");
      return false;
    }
    *out = &e;
    return true;
  }
  // bounded cache: a stream of differently sized images (FDDB) must not pile up device tables
  const size_t cap = (size_t)std::max<long long>(2, c->kn.plan_cache);
  while (c->plans.size() >= cap) {
    // least recently used plan that no submitted batch still runs on (PlanEntry::pins): a pending ticket's kernels
    // read the plan's device tables until its Wait
    auto victim = c->plans.end();
    for (auto p = c->plans.begin(); p != c->plans.end(); ++p)
      if (p->second.pins == 0 && (victim == c->plans.end() || p->second.last_use < victim->second.last_use)) victim = p;
    if (victim == c->plans.end()) break;        // every plan is in use: exceed the cap for now
    c->plan_pool.push_back({victim->second.dp, victim->second.table, victim->second.table_cap});
    c->plans.erase(victim);
  }
  if ((int)sp.levels.size() > kMaxLevels) { fail("too many pyramid levels"); return false; }
  if (!ragged && sp.windows * 1LL > 0x7fffffffLL) { fail("frame has too many windows"); return false; }
  if (sp.width > 65535 || sp.height > 65535) { fail("frames wider or taller than 65535 pixels are not supported"); return false; }
  PlanEntry pe;
  pe.sp = sp;
  // LDS-tiled stage-0 scan needs every stage-0 node to read the origin image
  bool s0_plain = true;
  const size_t n0 = (size_t)c->hm.K * c->hm.node_n();
  for (size_t i = 0; i < n0; i++) s0_plain = s0_plain && c->hm.nodes[i].scale == 0;
  pe.fast_scan = s0_plain && c->kn.no_fast_scan == 0;
  assign_tiles(sp, c->hm, c->kn, pe.fast_scan, dialect == JDA_DIALECT_C ? 4 : 8, &pe, ragged);
  size_t entries = 0;
  pe.lm_ok = true;
  for (int i = 0; i < pe.hp.n_levels; i++)
    if (pe.hp.lv[i].tiled) {
      entries += n0;
      if (pe.hp.lv[i].win > 2047) pe.lm_ok = false;      // (x, y) inside the window: 11 bits each
    }
  S0Node* stale_table = nullptr;
  if (!c->plan_pool.empty()) {            // recycle an evicted plan's allocations
    Cascador::PlanBuffers b = c->plan_pool.back();
    c->plan_pool.pop_back();
    pe.dp = b.dp; pe.table = b.table; pe.table_cap = b.table_cap;
    if (pe.table_cap < entries) { stale_table = pe.table; pe.table = nullptr; pe.table_cap = 0; }      // (too small: freed below, without the lock)
  }
  pe.last_use = ++c->plan_clock;
  pe.pred_tail = c->pred_tail; pe.pred_out = c->pred_out;      // a new frame size starts from the cascador's last pass
  pe.dense_hint = c->last_dense;
  pe.pins = 1;
  pe.building = true;
  const void* nodes = dialect == JDA_DIALECT_C ? c->mf.m.nodes : c->md.m.nodes;
  const void* ms = dialect == JDA_DIALECT_C ? (const void*)c->mf.m.mean_shape : (const void*)c->md.m.mean_shape;
  const int K = c->hm.K, node_n = c->hm.node_n();
  hipStream_t aux = c->aux;
  PlanEntry& e = c->plans.emplace(key, std::move(pe)).first->second;
  // ---- device side, without the lock: only this thread touches the entry's device pointers while `building` ----
  lk.unlock();
  // (a failure below must not lose the device allocations: whatever the entry holds goes back to the pool)
  if (stale_table) (void)hipFree(stale_table);
  auto build = [&](PlanEntry& pe) -> bool {
    if (!pe.dp) JDA_HIP(hipMalloc((void**)&pe.dp, sizeof(DevPlan)));
    JDA_HIP(hipMemcpy(pe.dp, &pe.hp, sizeof(DevPlan), hipMemcpyHostToDevice));
    if (entries) {
      if (!pe.table) {
        pe.table_cap = std::max(entries, (size_t)16 * n0);      // room for 16 levels: most recycled tables fit the next plan
        JDA_HIP(hipMalloc((void**)&pe.table, 2 * pe.table_cap * sizeof(S0Node)));   // cart-major tables + their level-major copy
      }
      JDA_HIP(launch_prep_stage0(dialect, pe.dp, pe.hp, nodes, ms, K, node_n, pe.table, pe.table + pe.table_cap, aux));
      // the scans that read the table run on the lanes' streams
      JDA_HIP(hipStreamSynchronize(aux));
    }
    return true;
  };
  const bool ok = build(e);
  lk.lock();
  e.building = false;
  if (!ok) {
    if (e.dp || e.table) c->plan_pool.push_back({e.dp, e.table, e.table ? e.table_cap : 0});
    e.dp = nullptr; e.table = nullptr; e.table_cap = 0;
    e.failed = true;
    if (--e.pins == 0) c->plans.erase(key);       // (callers that wait for this plan see `failed` and remove it when the last has left)
    c->plan_cv.notify_all();
    return false;
  }
  c->plan_cv.notify_all();
  *out = &e;
  return true;
}

void unpin_plan(Cascador* c, PlanEntry* pe) {
  if (!pe) return;
  std::lock_guard<std::mutex> lk(c->mu);
  if (pe->pins > 0) pe->pins--;
}

}  // namespace jda
