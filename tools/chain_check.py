"""Run-to-run differences of 10 replayed steps (small config) in four modes: one update graph (twice), two update graphs
unchained, two update graphs chained.  Prints max |difference| of losses / parameters / BatchNorm buffers against the first run."""
import contextlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from spadot_amd.model import SpaDOT
from spadot_amd.ops import FlatAdamW
from spadot_amd.synthetic import make_dataset
from spadot_amd.utils import _train_utils as tu, _utils
from test_train_gpu import _small_config
torch.randn_like = lambda x, **k: torch.zeros_like(x)
DEV = "cuda:0"
dt = torch.bfloat16 if "bf16" in sys.argv else torch.float32
data = make_dataset(2, 1200, 40, seed=3)
cfg = _small_config()
cfg.update(input_dim=40, timepoints=[0, 1], device=torch.device(DEV), staged_graphs=True, compute_dtype=dt)
res = []
for mode in ("one_graph", "one_graph", "unchained", "chained", "chained"):
    _utils.set_seed(7)
    cfg["split_update"] = mode != "one_graph"
    dd = tu.prepare_dataloader(data, cfg)
    model = SpaDOT.SpaDOT(cfg, dd).to(DEV)
    opt = FlatAdamW(model.parameters(), lr=cfg["lr"], first=model.SVGPEncoder.parameters())
    tu._update_Kmeans(model, cfg, dd); tu._update_OT_matrix(model, cfg)
    model.train()
    st = tu.GraphedStepper(model, opt, cfg, dd)
    losses = []
    for rep in range(5):
        with (st.chained() if mode == "chained" else contextlib.nullcontext()):
            for bi in range(2):
                losses.append(st.step(1, 1, bi, 5, 0.3).clone())
    torch.cuda.synchronize()
    bn = torch.cat([b.detach().double().reshape(-1) for b in model.SVGPEncoder.buffers()])
    res.append((torch.stack(losses).double(), opt.flat_param.detach().double().clone(), bn))
    l0, p0, b0 = res[0]
    l, p, b = res[-1]
    print(f"{mode:10s} losses {float((l - l0).abs().max()):.3e} (rel {float(((l - l0).abs() / (l0.abs() + 1e-9)).max()):.2e})  "
          f"params {float((p - p0).abs().max()):.3e}  bn {float((b - b0).abs().max()):.3e}", flush=True)
