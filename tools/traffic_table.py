"""Per-kernel HBM byte accounting of the many-workgroup MPM backward (where does shape_rope's backward traffic go).

Two inputs, both taken on the GPU box by tools/traffic_table.sh:
  * the per-kernel FETCH_SIZE / WRITE_SIZE of one step call (tools/pmc_large.sh -> gpurun_out/pmc_large_summary_<name>.csv; one env group,
    counters per launch, averaged over the launches of the pass);
  * the forward's own grid checkpoint of one step call, read back here: per substep the number of active cells of every env (the record
    index) and the cell keys (the record pool), from which the number of distinct 128-byte lines of a dense float4 grid those cells lie on
    follows -- tools/ubench_fetch.hip shows that on gfx950 every memory-side read request is 128 B (a scattered 16-B cell read costs a
    128-B line; FETCH_SIZE tallies it at 64: the x2 of MI355X_MICROARCH.md holds for streams and gathers alike) and that float atomics
    fetch nothing and are written as the dwords they carry.
From those the bytes each kernel moves BY DESIGN are written down buffer by buffer and compared with 2 x FETCH_SIZE and WRITE_SIZE.
usage (GPU box): python tools/traffic_table.py shape_rope gpurun_out/pmc_large_summary_shape_rope.csv"""
import csv
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def checkpoint_cells(name, B=32):
    """One simulator.step of the env's reset state with a checkpoint; returns (N, Np, S, n_prim, cells[S][B], lines128[S][B], svd_rows)."""
    from unidom_amd.engine.mpm_simulator import _Step
    from unidom_amd.envs.registration import env_functions
    from unidom_amd.utils import prng
    env = env_functions[name](batch_size=B, seed=0, device="cuda:0")
    _, st = env.reset(prng.PRNGKey(0))
    sim = env.simulator
    N, S, P = sim.n_particles, sim.conf.steps, sim.n_primitive
    res = tuple(int(r) for r in sim.conf.res)
    req = lambda t: t.detach().clone().requires_grad_(True)
    prim = st.primitives[0] if P == 1 else None
    assert prim is not None, "one primitive only (shape_rope, the scaled ropes)"
    act = torch.zeros((B, 6), device="cuda:0")
    act[:, 0] = 0.3
    # the env's focus shift (mpm_env.py:99-114): the body's mean (x, z) sits at the middle of the `res` box, the primitive moves along
    n_grid = float(sim.conf.n_grid)
    shift = torch.tensor([res[0] * 0.5 / n_grid, 0.0, res[2] * 0.5 / n_grid], device="cuda:0") - st.x.mean(1)
    shift[:, 1] = 0
    x0, p0 = st.x + shift[:, None], prim.position + shift[:, None]
    out = _Step.apply(sim, req(x0), req(st.v), req(st.C), req(st.F), st.J, req(p0), prim.rotation, prim.size, st.friction, st.mu, st.lamda, act)
    torch.cuda.synchronize()
    ckpt = out[0].grad_fn.saved_tensors[0]
    stride = ckpt.numel() // B
    Np = (N + 15) // 16 * 16
    gck = int(sim.grid_ckpt_cells)
    budget = S * gck * N
    # ck_layout (csrc/mpm_large.hip): history (S + 1) records of (24 [+ 21 SVD rows]) * Np | primitive tail n_prim * S * 10 | index (S + 1 ints, padded to 4)
    # | pool budget * 8 | [collide records] | order Np -- the SVD rows ride along in the four-lane regime (B * N < 100 000)
    for svd in (21, 0):
        rec = (24 + svd) * Np
        off_idx = ((S + 1) * rec + P * S * 10 + 3) // 4 * 4
        off_pool = off_idx + (S + 1 + 3) // 4 * 4
        if off_pool + budget * 8 <= stride:
            idx = ckpt.view(torch.int32).reshape(B, stride)[:, off_idx:off_idx + S + 1].cpu().numpy()
            if (idx[:, 0] == 0).all() and (np.diff(idx, axis=1) > 0).all() and idx[:, -1].max() <= budget:
                break
    else:
        raise SystemExit("checkpoint layout not recognised (ck_layout changed?)")
    pool = ckpt.view(torch.int32).reshape(B, stride)[:, off_pool:off_pool + budget * 8].reshape(B, budget, 8)[:, :, 0].cpu().numpy()
    cells = np.diff(idx, axis=1).T                                  # [S][B]
    if os.environ.get("TT_DEBUG"):
        # cross-check: the stencil cells of the history's positions at a few substeps of env 0, counted on the host
        hist = ckpt.reshape(B, stride)[0, :(S + 1) * rec].reshape(S + 1, 24 + svd, Np).cpu().numpy()
        n_grid = float(sim.conf.n_grid)
        for f in (0, 1, S // 2, S - 1):
            xs = hist[f, 0:3, :N].T
            base = (xs * n_grid - 0.5).astype(np.int32)
            cs = set()
            for bb in base:
                for i in range(3):
                    for j in range(3):
                        for kk in range(3):
                            c3 = (bb[0] + i, bb[1] + j, bb[2] + kk)
                            if all(0 <= c3[d] < res[d] for d in range(3)):
                                cs.add(c3)
            print(f"# debug env 0 substep {f}: records {cells[f, 0]}, host stencil cells inside res {len(cs)}, x range {xs.min(0)} .. {xs.max(0)}", file=sys.stderr)
        print("# debug idx env 0:", idx[0, :12], "...", idx[0, -3:], file=sys.stderr)
        print("# debug cells per env at substep 0:", cells[0].tolist(), file=sys.stderr)
        print("# debug cells per env at substep S-1:", cells[-1].tolist(), file=sys.stderr)
        xs0 = x0.detach().cpu().numpy()
        print("# debug x range per env (min xyz / max xyz) env 0, 1, 2:", [(xs0[b].min(0).round(3).tolist(), xs0[b].max(0).round(3).tolist()) for b in range(3)], "finite:", bool(np.isfinite(xs0).all()), file=sys.stderr)
    lines, sectors = np.zeros_like(cells), np.zeros_like(cells)
    for b in range(B):
        for f in range(S):
            key = pool[b, idx[b, f]:idx[b, f + 1]]
            lin = ((key & 1023).astype(np.int64) * res[1] + ((key >> 10) & 1023)) * res[2] + ((key >> 20) & 1023)
            lines[f, b] = np.unique(lin >> 3).size                 # 8 float4 cells per 128-B line (read granularity)
            sectors[f, b] = np.unique(lin >> 1).size               # 2 float4 cells per 32-B sector (write granularity: tools/ubench_fetch.hip scatter16)
    # cells each 128-lane block of the particle kernel (32 consecutive slots of the spatial order) flushes from its LDS table: the distinct
    # (clamped) stencil cells of its particles, sampled at a few substeps from the history's positions
    hist_x = ckpt.reshape(B, stride)[:, :(S + 1) * rec].reshape(B, S + 1, 24 + svd, Np)[:, :, 0:3, :N].cpu().numpy()
    n_grid = np.float32(sim.conf.n_grid)
    off = np.array([(i, j, k) for i in range(3) for j in range(3) for k in range(3)], np.int64)
    flush = []
    for f in range(0, S, max(1, S // 8)):
        tot = 0
        for b in range(B):
            base = (hist_x[b, f].T * n_grid - np.float32(0.5)).astype(np.int32).astype(np.int64)        # [N][3]
            c3 = np.clip(base[:, None, :] + off[None], 0, np.array(res) - 1)                              # gather clamp
            lin = (c3[..., 0] * res[1] + c3[..., 1]) * res[2] + c3[..., 2]                                 # [N][27]
            for p0 in range(0, N, 32):
                tot += np.unique(lin[p0:p0 + 32]).size
        flush.append(tot)
    row_bytes = float(np.mean([(((b * stride + r * Np + N) * 4 + 127) // 128 - ((b * stride + r * Np) * 4) // 128) * 128 for b in range(B) for r in range(24)]))
    return N, Np, S, P, cells, lines, svd, sectors, float(np.mean(flush)), row_bytes


def main():
    name, summary = sys.argv[1], sys.argv[2]
    B = 32
    N, Np, S, P, cells, lines, svd, sectors, flush_cells, row_bytes = checkpoint_cells(name, B)
    C, L = float(cells.sum(1).mean()), float(lines.sum(1).mean())   # per launch (all envs), mean over the substeps
    S32 = float(sectors.sum(1).mean())
    Pn = B * N
    rowB = row_bytes * B                                            # bytes one SoA row of all envs costs: N * 4 B per env rounded out to the 128-B lines it straddles
    k = {r["kernel"]: r for r in csv.DictReader(open(summary))}
    KB = 1024.0
    print(f"# {name}: {B} envs x {N} particles (rows of {Np}), {S} substeps per step call; active cells per env and substep {C / B:.0f} "
          f"(min {cells.min()}, max {cells.max()}), on {L / B:.0f} distinct 128-B lines of a float4 grid ({C / L:.2f} cells per line of 8)")
    print("# bytes per LAUNCH (all envs); counters: tools/pmc_large.sh (mean over the launches of a pass); fetch = 2 x FETCH_SIZE (tools/ubench_fetch.hip)")
    rows = []
    if "lg_gadj_restore" in k:
        # K1(f): blocks [0, nb) grid-op adjoint of substep f, blocks [nb, 2 nb) restore of substep f - 1
        rd = [("restore: checkpoint records (key, m, mv, v) 32 B / cell", 32 * C, "stream"),
              ("restore: cell list of substep f + 2 (zeroing its cotangent cells) 4 B / cell", 4 * C, "stream"),
              ("grid-op adjoint: cell list 4 B + record (m, mv) 32 B / cell", 36 * C, "stream"),
              ("grid-op adjoint: cotangent cells 16 B / cell, one 128-B line per touched line", 128 * L, "gather")]
        wr = [("restore: cell list 4 B / cell", 4 * C, "stream"),
              ("restore: velocity cells, 16 B / cell written in 32-B sectors (scattered)", 32 * S32, "scatter"),
              ("restore: cotangent cells of substep f + 2 zeroed, 32-B sectors (scattered)", 32 * S32, "scatter"),
              ("grid-op adjoint: cotangent cells, 32-B sectors (scattered)", 32 * S32, "scatter")]
        rows.append(("lg_gadj_restore", rd, wr))
    if "lg_padj_gadj" in k:
        n_rows = 24 + svd + 12 + 3 + 3 + 1                   # state rows, SVD rows, cotangent rows x / F, fx cotangent, x of substep f - 1, spatial order
        rd = [(f"p2g + particle adjoint: {n_rows} SoA rows per env (state 24, SVD factors {svd}, cotangent x / F 12, fx cotangent 3, x of f - 1 3, order 1), {N} x 4 B each in whole 128-B lines", n_rows * rowB, "stream"),
              ("p2g adjoint: cotangent cells of substep f gathered 27 / particle: one 128-B line per touched line", 128 * L, "gather"),
              ("g2p adjoint of f - 1: velocity cells gathered 27 / particle: one 128-B line per touched line", 128 * L, "gather"),
              (f"forward-kinematics adjoint block: the primitive's whole trajectory every substep ({S} rows x 3 floats of position, input position, both cotangent arrays: the reference clips the WHOLE array each substep, primitives.py:187)", B * P * 4 * S * 3 * 4, "stream")]
        wr = [("cotangent rows x / F 12 + fx cotangent 3 (4 B each)", 4 * 15 * Pn, "stream"),
              ("forward-kinematics adjoint block: both cotangent trajectories rewritten", B * P * 2 * S * 3 * 4, "stream"),
              (f"g2p adjoint of f - 1: float atomics flushing each block's LDS table, 3 dwords per cell and block ({flush_cells / B:.0f} (block, cell) pairs per env)", 12 * flush_cells, "atomic")]
        rows.append(("lg_padj_gadj", rd, wr))
    tot_d = tot_c = 0.0
    for kn, rd, wr in rows:
        f2, w = 2 * float(k[kn]["FETCH_SIZE"]) * KB, float(k[kn]["WRITE_SIZE"]) * KB
        dr, dw = sum(x[1] for x in rd), sum(x[1] for x in wr)
        print(f"\n{kn}: {int(float(k[kn]['launches']))} launches in the pass")
        for what, by, kind in rd:
            print(f"   read   {by / 1e6:8.3f} MB  [{kind}]  {what}")
        print(f"   read   {dr / 1e6:8.3f} MB designed   vs   {f2 / 1e6:8.3f} MB = 2 x FETCH_SIZE   ({100 * dr / f2:.0f} % accounted)")
        for what, by, kind in wr:
            print(f"   write  {by / 1e6:8.3f} MB  [{kind}]  {what}")
        print(f"   write  {dw / 1e6:8.3f} MB designed   vs   {w / 1e6:8.3f} MB = WRITE_SIZE       ({100 * dw / w:.0f} % accounted)")
        tot_d += dr + dw
        tot_c += f2 + w
    alg = B * (288 * N + 112 * C / B)
    print(f"\nper reverse substep (both launches): designed {tot_d / 1e6:.2f} MB, counters {tot_c / 1e6:.2f} MB ({100 * tot_d / tot_c:.0f} % accounted); "
          f"SURVEY 8(d) algorithmic 288 N + 112 G_act = {alg / 1e6:.2f} MB -> counters / algorithmic = {tot_c / alg:.2f}")
    over = [("SVD factor rows read back instead of iterated again", svd * rowB),
            ("grid-checkpoint records: restore + grid-op adjoint read them where the algorithmic count has one 16-B grid read each", (32 + 4 + 36) * C - 2 * 16 * C),
            ("128-B line granularity of the three grid gathers against 16 B per touched cell", 3 * 128 * L - 3 * 16 * C),
            ("spatial order, fx cotangent scratch written and read back (4-B rows), rows rounded out to 128-B lines", 7 * rowB + 24 * (rowB - 4 * Pn)),
            ("32-B sector granularity of the scattered cell writes and the per-block atomics beyond one per cell", 3 * (32 * S32 - 16 * C) + 12 * (flush_cells - C))]
    print(f"not itemised ({100 - 100 * tot_d / tot_c:.0f} % of the counter bytes): kernel arguments and code, per-env scalars and index words, the primitive rows every lane of the grid-op "
          "adjoint loads, lines evicted and fetched again inside a launch")
    print("what the counters hold beyond the algorithmic count, by design:")
    for what, by in over:
        print(f"   {by / 1e6:7.3f} MB  {what}")


if __name__ == "__main__":
    main()
