"""CPU simulation of the LDS bank conflicts of k_scan's pixel gathers (no GPU): how many LDS passes the two byte reads
per tree node cost a wave under (A) the product's lane assignment -- regular mapping before the first compaction, then
order-preserving compaction -- and (B) a bank-aware order of every compacted queue (survivors dealt round-robin over the
32 banks of their window origins, so that the 32 lanes of a half-wave start from 32 different banks).
Cost model of one 64-lane ds_read_u8 (tools/lds_bench.hip agrees with it: 2.6 / 4.4 / 8.1 clocks for consecutive /
neighbour-stride / random addresses): two halves of 32 lanes; a half takes as many passes as its busiest bank has
DISTINCT dwords among the active lanes.
usage: python tools/experiments/bank_sim.py [frames]
"""
import sys
import numpy as np

sys.path.insert(0, ".")
from jda_amd import synth  # noqa: E402

f32 = np.float32
PHASES = [(0, 16), (16, 32), (32, 64), (64, 128)]


def instr_cost(dwords, active):
    """dwords [64] int, active [64] bool -> passes"""
    c = 0
    for h in (slice(0, 32), slice(32, 64)):
        d = np.unique(dwords[h][active[h]])
        if len(d):
            c += np.bincount(d & 31, minlength=32).max()
    return c


def level_tables(m, win):
    """stage-0 pixel offsets per (cart, node): x1,y1,x2,y2 (c/jda.c:373-389 with the mean shape)"""
    ms = m.mean_shape.astype(f32)
    off = m.off[0].astype(f32)                      # [K, node, 4]
    l1, l2 = m.lm1[0], m.lm2[0]
    def coord(v):
        return np.clip(np.trunc((v * f32(win)).astype(f32)).astype(np.int64), 0, win - 1)
    x1 = coord(ms[2 * l1] + off[..., 0]); y1 = coord(ms[2 * l1 + 1] + off[..., 1])
    x2 = coord(ms[2 * l2] + off[..., 2]); y2 = coord(ms[2 * l2 + 1] + off[..., 3])
    return x1, y1, x2, y2


def simulate(m, frames, win, tw, th, bank_aware_from=1, static_perm=False, K=128):
    H, W = frames.shape[1:]
    step = int(win * 0.1)
    nx, ny = (W - win) // step + 1, (H - win) // step + 1
    pw = win + (tw - 1) * step
    pitch = (pw + 15) & ~15
    if pitch % 128 == 0:
        pitch += 16
    x1, y1, x2, y2 = level_tables(m, win)
    leaf = m.leaf[0].astype(f32); cth = m.cth[0].astype(f32); cmean = m.cmean[0].astype(f32); cstd = m.cstd[0].astype(f32)
    nth = m.nth[0]
    tot = {"A": np.zeros(len(PHASES)), "B": np.zeros(len(PHASES))}
    instrs = {"A": np.zeros(len(PHASES)), "B": np.zeros(len(PHASES))}
    lanes_used = np.zeros(len(PHASES))
    for fr in frames:
        for ty in range((ny + th - 1) // th):
            for tx in range((nx + tw - 1) // tw):
                n = tw * th
                i, j = np.arange(n) % tw, np.arange(n) // tw
                real = (tx * tw + i < nx) & (ty * th + j < ny)
                X0, Y0 = (tx * tw + i) * step, (ty * th + j) * step
                X0 = np.where(real, X0, 0); Y0 = np.where(real, Y0, 0)
                origin = (j * step) * pitch + i * step            # tile-relative byte address of the window origin
                # walk all K carts for every window; record the two byte addresses per depth, and the reject cart
                addr = np.zeros((n, K, 3, 2), np.int64)
                score = np.zeros(n, f32)
                dead_at = np.full(n, K, np.int64)                 # first cart index after which the window is dead (K = survives)
                alive = real.copy()
                for k in range(K):
                    at = np.zeros(n, np.int64)
                    for d in range(3):
                        a1 = origin + y1[k][at] * pitch + x1[k][at]
                        a2 = origin + y2[k][at] * pitch + x2[k][at]
                        addr[:, k, d, 0], addr[:, k, d, 1] = a1, a2
                        p1 = fr[Y0 + y1[k][at], X0 + x1[k][at]].astype(np.int64)
                        p2 = fr[Y0 + y2[k][at], X0 + x2[k][at]].astype(np.int64)
                        at = 2 * at + np.where(p1 - p2 <= nth[k][at], 1, 2)
                    score = ((score + leaf[k][at - 7] - cmean[k]) / cstd[k]).astype(f32)
                    rej = alive & (score < cth[k])
                    dead_at[rej] = k
                    alive &= ~rej
                dead_at[~real] = -1
                for scheme in ("A", "B"):
                    queue = np.arange(n)
                    if scheme == "B" and static_perm:
                        queue = bank_order(queue, origin)
                    for pi, (k0, k1) in enumerate(PHASES):
                        if pi > 0:
                            queue = queue[dead_at[queue] >= k0]       # survivors of every cart before k0, order kept
                            if scheme == "B" and pi >= bank_aware_from:
                                queue = bank_order(queue, origin)
                        if scheme == "A":
                            lanes_used[pi] += len(queue)
                        for w0 in range(0, len(queue), 64):
                            lanes = queue[w0:w0 + 64]
                            pad = 64 - len(lanes)
                            for kb in range(k0, k1, 4):               # four carts in flight; reject applied after the four
                                act = dead_at[lanes] >= kb
                                if not act.any():
                                    continue
                                act64 = np.concatenate([act, np.zeros(pad, bool)])
                                for k in range(kb, kb + 4):
                                    for d in range(3):
                                        for p in range(2):
                                            dw = np.concatenate([addr[lanes, k, d, p] >> 2, np.zeros(pad, np.int64)])
                                            tot[scheme][pi] += instr_cost(dw, act64)
                                            instrs[scheme][pi] += 1
    return tot, instrs, lanes_used


def bank_order(queue, origin):
    """deal the queue's windows round-robin over the 32 banks of their origins: sort by (rank within bank, bank)"""
    if len(queue) == 0:
        return queue
    b = (origin[queue] >> 2) & 31
    order = np.argsort(b, kind="stable")
    bs = b[order]
    start = np.searchsorted(bs, np.arange(32))
    rank = np.arange(len(queue)) - start[bs]
    key = rank * 32 + bs
    return queue[order][np.argsort(key, kind="stable")]


if __name__ == "__main__":
    nf = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    m = synth.make_model(5, 540, 27, 4, seed=1)
    calib = synth.make_frames(8, 640, 480, seed=0, first=10_000_000)
    synth.calibrate_thresholds(m, calib)
    frames = synth.make_frames(nf, 640, 480, seed=0)
    for win, tw, th in ((46, 50, 10), (57, 17, 22), (71, 21, 20), (88, 14, 25)):
        for static in (False, True):
            tot, ins, used = simulate(m, frames, win, tw, th, static_perm=static)
            a, b = tot["A"], tot["B"]
            print("win %3d tile %dx%d step %d  %s" % (win, tw, th, int(win * 0.1), "B also permutes phase 0" if static else "B reorders compacted queues"))
            for pi, ph in enumerate(PHASES):
                print("   carts [%3d,%3d): instr %8d  passes A %9d (%.2f/instr)  B %9d (%.2f/instr)  lanes %d"
                      % (ph[0], ph[1], ins["A"][pi], a[pi], a[pi] / max(ins["A"][pi], 1), b[pi], b[pi] / max(ins["B"][pi], 1), used[pi]))
            print("   total passes A %d  B %d  -> B/A %.3f" % (a.sum(), b.sum(), b.sum() / a.sum()))
            sys.stdout.flush()
