import sys; sys.path.insert(0, ".")
import torch
from spadot_amd import ops
DEV="cuda"
w = torch.tensor([0.1, -0.4, 1e-4, 0.1, 0.1, 1.0], device=DEV)
terms = [torch.tensor(v, device=DEV, requires_grad=(i != 4)) for i, v in enumerate([3.0, -2.0, 50.0, 0.25, 7.0, 0.5])]
print([hex(t.data_ptr()) for t in terms], [t.dtype for t in terms])
elbo, log7 = ops.mix_losses(w, terms)
torch.cuda.synchronize()
print(elbo, log7)
