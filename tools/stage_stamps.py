"""Where the stages of a replayed training step really start and end on the GPU, with NO profiler attached (VERDICT r03 item 3:
under rocprofv3 the SVGP backward starts 230-250 us after its input exists -- is that the profiler or the schedule?).

    SPADOT_STAMPS=1 python tools/stage_stamps.py [--steps 40]

GraphedStepper (SPADOT_STAMPS=1) captures a one-thread timestamp launch at the head and the end of every stage graph
(csrc: k_stamp, the device's 100 MHz counter); this script runs chained cfg3 steps, reads the 16 stamps after each step and
prints the median timeline relative to the start of the GAT forward."""
import argparse
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SPADOT_STAMPS"] = "1"

from spadot_amd.model import SpaDOT  # noqa: E402
from spadot_amd.ops import FlatAdamW  # noqa: E402
from spadot_amd.synthetic import make_dataset  # noqa: E402
from spadot_amd.utils import _train_utils as tu, _utils  # noqa: E402

NAMES = {0: "gat_fwd", 1: "svgp_fwd", 2: "tail", 3: "svgp_bwd", 4: "gat_bwd_a", 5: "gat_bwd_b", 6: "late", 7: "svgp_pre", 12: "upd_head", 13: "upd_rest"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--spots", type=int, default=10000)
    ap.add_argument("--genes", type=int, default=3000)
    args = ap.parse_args()
    dev = "cuda:0"
    T = 2
    data = make_dataset(T, args.spots, args.genes, seed=1993)
    cfg = _utils.load_model_config(types.SimpleNamespace(config=None))
    cfg.update(input_dim=args.genes, timepoints=list(range(T)), device=torch.device(dev), compute_dtype=torch.bfloat16,
               inducing_point_nums=480)
    _utils.set_seed(cfg["seed"])
    dd = tu.prepare_dataloader(data, cfg)
    model = SpaDOT.SpaDOT(cfg, dd).to(dev)
    opt = FlatAdamW(model.parameters(), lr=cfg["lr"], last=model.GATEncoder.first_layer_parameters(), first=model.SVGPEncoder.parameters())
    tu._update_Kmeans(model, cfg, dd)
    tu._update_OT_matrix(model, cfg)
    model.train()
    st = tu.GraphedStepper(model, opt, cfg, dd)
    ep = cfg["ot_epoch"]
    nb = len(dd["dataloaders"][1])
    for rep in range(3):
        for bi in range(nb):
            st.step(1, 1, bi, ep, 0.5)
    torch.cuda.synchronize()
    rows = []
    # two chained steps, then one read-out: every stamp then belongs to the SECOND step (a read-out synchronises, so the
    # first step of each pair starts unchained; the second one is the steady-state step)
    with st.chained():
        for k in range(args.steps):
            st.step(1, 1, k % nb, ep, 0.5)
            st.step(1, 1, (k + 1) % nb, ep, 0.5)
            torch.cuda.synchronize()
            rows.append(st.stamps.cpu().numpy().astype(np.int64).copy())
    a = np.stack(rows).astype(np.float64) * 0.01          # 10 ns ticks -> microseconds
    t0 = a[:, 0:1]
    rel = a - t0
    med = np.median(rel, axis=0)
    print(f"median over {len(rows)} replayed steps (us, relative to the start of the GAT forward graph):")
    for k, name in sorted(NAMES.items()):
        s, e = med[2 * k], med[2 * k + 1]
        if a[:, 2 * k].max() == 0:
            continue
        print(f"  {name:10s} start {s:9.1f}  end {e:9.1f}  ({e - s:7.1f})")
    sv_start, sv_end = med[2], med[3]
    if a[:, 16].max() > 0:
        print(f"  inside svgp_fwd: encoder {med[16] - med[2]:.1f} us, Sigma build {med[17] - med[16]:.1f}, inverse {med[18] - med[17]:.1f}, "
              f"behind the inverse {med[3] - med[18]:.1f}")
    if a[:, 19].max() > 0 and a[:, 28].max() > 0:
        print(f"  inside the encoder: first map {med[19] - med[2]:.1f} us, BN {med[20] - med[19]:.1f}, hidden map {med[21] - med[20]:.1f}, "
              f"BN {med[22] - med[21]:.1f}, SVGP_fc {med[16] - med[22]:.1f}; Sigma build: pre2 {med[23] - med[16]:.1f}, G {med[28] - med[23]:.1f}, "
              f"t {med[17] - med[28]:.1f}")
    print(f"  svgp_bwd starts {med[6] - med[5]:.1f} us after the tail ends; gat_bwd_a starts {med[8] - med[5]:.1f} us after the tail ends")
    print(f"  svgp_fwd ends {sv_end - med[1]:+.1f} us relative to gat_fwd's end; the queue (late) ends {med[13] - med[11]:+.1f} us relative to gat_bwd_b's end")
    print(f"  step period (upd_rest end of this step - upd_rest end of the previous one is not stamped): gat_fwd start -> upd_rest end {med[27]:.1f} us")


if __name__ == "__main__":
    main()
