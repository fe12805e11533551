"""Decoder half of the VAE: latent sample [b, z_dim] -> reconstructed expression [b, G].

Same module tree as the reference's decoder (/root/reference/SpaDOT/model/decoder.py:3-20), because the
state_dict keys are part of the drop-in surface: `decoder_net` is a Sequential whose entries 0/3/6 are the
dense maps, 1/4 the LayerNorms, 2/5 the LeakyReLUs (hidden maps Xavier-uniform, output map default init).
The dense maps run on MFMA through the library GEMM; LayerNorm / LeakyReLU are device element-wise kernels.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..ops import (grad_bias, linear_bias, ln_act, mlp_chain, mlp_chain_ok, recon_fb_ok, recon_sqerr, recon_sqerr_fb, recon_sqerr_ok,
                   sqerr_sum, weight_image)


def _hidden_stage(fan_in, fan_out):
    dense = nn.Linear(fan_in, fan_out)
    nn.init.xavier_uniform_(dense.weight)
    return dense, nn.LayerNorm(fan_out), nn.LeakyReLU()


class Decoder(nn.Module):
    def __init__(self, input_dim, z_dim, decoder_layers, compute_dtype=torch.float32):
        super().__init__()
        self.compute_dtype = compute_dtype
        widths = [z_dim, *decoder_layers]
        stages = []
        for fan_in, fan_out in zip(widths[:-1], widths[1:]):
            stages.extend(_hidden_stage(fan_in, fan_out))
        stages.append(nn.Linear(widths[-1], input_dim))
        self.decoder_net = nn.Sequential(*stages)

    def forward(self, latent_sample):
        h, last, _ = self._hidden(latent_sample)
        # hidden -> G is the only large GEMM here: compute dtype on MFMA (fp32 accumulate), fp32 result
        return linear_bias(h, last.weight, last.bias, self.compute_dtype)

    def recon_loss(self, latent_sample, y, inv_scale, dz_extra=None, grad_weight=None):
        """inv_scale * sum (y - decoder(latent))^2 (SpaDOT.py:89) without materialising the reconstruction separately: in the
        bf16 compute dtype the bias add, the squared error and its sum are one launch behind the output map's GEMM.
        grad_weight (a device scalar, optional): the caller PROMISES that the backward pass will be seeded with exactly this
        tensor as d loss / d recon (GraphedStepper's tail: the loss weight lambda1 through mix_losses); the output map, the
        term and their backward are then ONE launch (ops.recon_sqerr_fb) -- which refuses any other seed."""
        bf = self.compute_dtype == torch.bfloat16
        nocast = bf
        # dz_extra: a gradient for latent_sample that arrives by no backward path of its own (ops.cluster_losses_fb); the hidden
        # stages' backward launch adds it
        h, last, hb = self._hidden(latent_sample, bf16_out=nocast, dx_add=dz_extra)
        if bf and grad_weight is not None:
            wim = self._output_image(last.weight)
            wimT = self._output_image_T(wim) if wim is not None else None
            if recon_fb_ok(h, last.weight, last.bias, y, hb, wim, wimT, grad_weight):
                return recon_sqerr_fb(h, last.weight, last.bias, y, inv_scale, hb, wim, wimT, grad_weight)
        if bf and recon_sqerr_ok(h, last.weight, last.bias, y):
            # no cast launch between the hidden stages and the output map: the chain's launch leaves a bf16 copy of its result,
            # and under an optimizer that keeps bf16 weight images current the map's weight needs none either
            return recon_sqerr(h, last.weight, last.bias, y, inv_scale, hb, self._output_image(last.weight) if nocast else None)
        return sqerr_sum(y, linear_bias(h, last.weight, last.bias, self.compute_dtype), inv_scale)

    def _output_image(self, W):
        """The bf16 image of the output map's weight that a FlatAdamW pinned by GraphedStepper keeps current (its update kernel
        writes it), or None: not training, no such optimizer, or one that no longer owns W (same rules as GATEncoder)."""
        opt = getattr(self, "_image_optimizer", None)
        if opt is None or not self.training or not W.is_cuda:
            return None
        if not opt.owns(W):
            object.__setattr__(self, "_image_optimizer", None)
            return None
        im = weight_image(W, W.shape[1], torch.bfloat16, self, tag="_wout")
        if not opt.maintain_image(W, im):
            return None
        opt.sync_images()
        return im

    def _output_image_T(self, wim, refresh=None):
        """The TRANSPOSED bf16 image [K, G rounded up to 128] of the output map's weight (ops.recon_sqerr_fb reads it for the
        input gradient).  Brought up to date from `wim` (the image the optimizer keeps current) by one strided copy: at the
        head of a step when a stepper calls step_begin() -- the main stream has slack there -- else right here."""
        G, K = wim.shape
        Gp = (G + 127) // 128 * 128
        t = getattr(self, "_wout_T", None)
        if t is None or t.shape != (K, Gp) or t.device != wim.device:
            t = torch.zeros((K, Gp), dtype=torch.bfloat16, device=wim.device)
            object.__setattr__(self, "_wout_T", t)
            object.__setattr__(self, "_wout_T_fresh", False)
        if refresh is True or (refresh is None and not getattr(self, "_wout_T_fresh", False)):
            with torch.no_grad():
                t[:, :G].copy_(wim.t())
        object.__setattr__(self, "_wout_T_fresh", bool(refresh))          # (consumed by the step's recon_loss call)
        return t

    def step_begin(self):
        """Called by GraphedStepper at the head of a step (inside its first captured stage): refreshes what the loss tail reads
        besides the optimizer-maintained images -- the transposed output-map image -- where the main stream has slack."""
        if self.compute_dtype != torch.bfloat16 or not self.training:
            return
        wim = self._output_image(list(self.decoder_net)[-1].weight)
        if wim is not None:
            self._output_image_T(wim, refresh=True)

    def _hidden(self, latent_sample, bf16_out=False, dx_add=None):
        stages = list(self.decoder_net)
        h, hb = latent_sample, None
        hidden = [(stages[i], stages[i + 1], stages[i + 2].negative_slope) for i in range(0, len(stages) - 1, 3)]
        if hidden and mlp_chain_ok(h, hidden):
            if bf16_out:
                h, hb = mlp_chain(h, hidden, bf16_out=True, dx_add=dx_add)     # all hidden stages: one launch forward, two backward
            else:
                h = mlp_chain(h, hidden, dx_add=dx_add)
        else:
            if dx_add is not None:
                h = grad_bias(h, dx_add)
            for dense, norm, slope in hidden:
                h = ln_act(linear_bias(h, dense.weight, dense.bias), norm, slope)     # LN + LeakyReLU: one launch
        return h, stages[-1], hb
