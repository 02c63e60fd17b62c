"""diagnostic: HIP fp32 gradients vs the oracle in fp64 and fp32 at MuST-C width"""
import copy, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch
from test_hip_config_width import *
from oracle import s2t_oracle as O

def run64(sd, ocfg, names, batch, ctc_w):
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    src, lengths, trg, tlen = batch
    return oracle_loss_and_grads(sd64, ocfg, names, src.double(), lengths, trg, tlen, ctc_w)

cfg = width_cfg(8, 2, 1, "xavier_normal"); V = 5000
torch.manual_seed(11)
base = make_model(cfg, V, None, None, None, 0.1)
sd = {k: v.clone() for k, v in base.state_dict().items()}
names = {n for n, _ in base.named_parameters()}
batch = synth_batch(V, [1498, 1203, 899], [60, 41, 72], seed=7)
ocfg = copy.deepcopy(cfg); ocfg["encoder"]["alpha"], ocfg["decoder"]["alpha"] = MUSTC_ALPHA
t0 = time.time(); l32, g32 = oracle_loss_and_grads(sd, ocfg, names, *batch, 0.1); print("oracle32", time.time() - t0, l32)
t0 = time.time(); l64, g64 = run64(sd, ocfg, names, batch, 0.1); print("oracle64", time.time() - t0, l64)
dev = torch.device("cuda:0") if torch.cuda.is_available() else None
gh = None
if dev is not None:
    model = make_model(cfg, V, sd, dev, torch.float32, 0.1, alpha=MUSTC_ALPHA)
    total, xent, ctc, ncor = model(return_type="loss", **vars(hip_batch(*batch, dev)))
    total.backward(); print("hip", total.item(), xent.item(), ctc.item())
    gh = {n: p.grad.cpu() for n, p in model.named_parameters()}
rows = []
for n in sorted(names):
    r = g64[n]; sc = r.abs().max().item() + 1e-12
    e32 = (g32[n].double() - r).abs().max().item() / sc
    eh = (gh[n].double() - r).abs().max().item() / sc if gh else float("nan")
    rows.append((eh, e32, n, sc))
rows.sort(reverse=True)
for eh, e32, n, sc in rows[:25]:
    print(f"{n:60s} hip {eh:.2e}  oracle32 {e32:.2e}  scale {sc:.3e}")
