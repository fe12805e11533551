"""Decoder half of the VAE: latent sample [b, z_dim] -> reconstructed expression [b, G].

Same module tree as the reference's decoder (/root/reference/SpaDOT/model/decoder.py:3-20), because the
state_dict keys are part of the drop-in surface: `decoder_net` is a Sequential whose entries 0/3/6 are the
dense maps, 1/4 the LayerNorms, 2/5 the LeakyReLUs (hidden maps Xavier-uniform, output map default init).
The dense maps run on MFMA through the library GEMM; LayerNorm / LeakyReLU are device element-wise kernels.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..ops import linear_bias, ln_act, mlp_chain, mlp_chain_ok, recon_sqerr, recon_sqerr_ok, sqerr_sum


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
        h, last = self._hidden(latent_sample)
        # hidden -> G is the only large GEMM here: compute dtype on MFMA (fp32 accumulate), fp32 result
        return linear_bias(h, last.weight, last.bias, self.compute_dtype)

    def recon_loss(self, latent_sample, y, inv_scale):
        """inv_scale * sum (y - decoder(latent))^2 (SpaDOT.py:89) without materialising the reconstruction separately: in the
        bf16 compute dtype the bias add, the squared error and its sum are one launch behind the output map's GEMM."""
        h, last = self._hidden(latent_sample)
        if self.compute_dtype == torch.bfloat16 and recon_sqerr_ok(h, last.weight, last.bias, y):
            return recon_sqerr(h, last.weight, last.bias, y, inv_scale)
        return sqerr_sum(y, linear_bias(h, last.weight, last.bias, self.compute_dtype), inv_scale)

    def _hidden(self, latent_sample):
        stages = list(self.decoder_net)
        h = latent_sample
        hidden = [(stages[i], stages[i + 1], stages[i + 2].negative_slope) for i in range(0, len(stages) - 1, 3)]
        if hidden and mlp_chain_ok(h, hidden):
            h = mlp_chain(h, hidden)                    # all hidden stages: one launch forward, two backward
        else:
            for dense, norm, slope in hidden:
                h = ln_act(linear_bias(h, dense.weight, dense.bias), norm, slope)     # LN + LeakyReLU: one launch
        return h, stages[-1]
