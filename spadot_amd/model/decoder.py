"""Decoder of the VAE: mirror of /root/reference/SpaDOT/model/decoder.py:3-20 (same module layout, hence
the same state_dict keys decoder_net.{0,1,3,4,6}.*).  Dense linears run on MFMA through the library
GEMM; LayerNorm/LeakyReLU are torch's device kernels."""
import torch.nn as nn


class Decoder(nn.Module):
    def __init__(self, input_dim, z_dim, decoder_layers):
        super().__init__()
        layers = [z_dim] + list(decoder_layers) + [input_dim]
        net = []
        for i in range(1, len(layers) - 1):
            lin = nn.Linear(layers[i - 1], layers[i])
            nn.init.xavier_uniform_(lin.weight)
            net += [lin, nn.LayerNorm(layers[i]), nn.LeakyReLU()]
        net.append(nn.Linear(layers[-2], layers[-1]))
        self.decoder_net = nn.Sequential(*net)

    def forward(self, latent_sample):
        return self.decoder_net(latent_sample)
