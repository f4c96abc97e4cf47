"""Reference-free deterministic parameter initialiser (TEST INFRASTRUCTURE ONLY).

Golden fixtures at full model size store only a seed: weights are regenerated on
any machine from numpy's MT19937 ``RandomState`` (bit-reproducible across
platforms), so 129 MB of weights never have to be committed.
Shapes follow SURVEY.md §8a (reference ``state_dict`` keys; models.py:109-165).
"""
import zlib
import numpy as np


def molvae_shapes(i=120, o=292, c=35, emb=30, h_enc=72, n_enc=3, h_dec=1024, n_dec=4):
    s = {}
    s["encoder.embedding.weight"] = (c, emb)
    for l in range(n_enc):
        s[f"encoder.gru.weight_ih_l{l}"] = (4 * h_enc, emb if l == 0 else h_enc)
        s[f"encoder.gru.weight_hh_l{l}"] = (4 * h_enc, h_enc)
        s[f"encoder.gru.bias_ih_l{l}"] = (4 * h_enc,)
        s[f"encoder.gru.bias_hh_l{l}"] = (4 * h_enc,)
    s["encoder.conv_1.0.weight"] = (120, i, 18); s["encoder.conv_1.0.bias"] = (120,)
    s["encoder.conv_2.0.weight"] = (64, 120, 18); s["encoder.conv_2.0.bias"] = (64,)
    s["encoder.conv_3.0.weight"] = (64, 64, 18); s["encoder.conv_3.0.bias"] = (64,)
    flat = (h_enc - 18 * 3 + 3) * 64
    s["encoder.dense_1.0.weight"] = (512, flat); s["encoder.dense_1.0.bias"] = (512,)
    s["encoder.lmbd.z_mean.weight"] = (o, 512); s["encoder.lmbd.z_mean.bias"] = (o,)
    s["encoder.lmbd.z_log_var.weight"] = (o, 512); s["encoder.lmbd.z_log_var.bias"] = (o,)
    s["decoder.latent_input.0.weight"] = (o, o); s["decoder.latent_input.0.bias"] = (o,)
    for l in range(n_dec):
        s[f"decoder.gru.weight_ih_l{l}"] = (4 * h_dec, o if l == 0 else h_dec)
        s[f"decoder.gru.weight_hh_l{l}"] = (4 * h_dec, h_dec)
        s[f"decoder.gru.bias_ih_l{l}"] = (4 * h_dec,)
        s[f"decoder.gru.bias_hh_l{l}"] = (4 * h_dec,)
    s["decoder.decoded_mean.module.0.weight"] = (c, h_dec)
    s["decoder.decoded_mean.module.0.bias"] = (c,)
    return s


def moses_shapes(V, q_h=256, d_z=160, d_h=512, n_dec=3):
    s = {"x_emb.weight": (V, V)}
    s["encoder_rnn.weight_ih_l0"] = (3 * q_h, V); s["encoder_rnn.weight_hh_l0"] = (3 * q_h, q_h)
    s["encoder_rnn.bias_ih_l0"] = (3 * q_h,); s["encoder_rnn.bias_hh_l0"] = (3 * q_h,)
    for n in ("q_mu", "q_logvar"):
        s[f"{n}.0.weight"] = (256, q_h); s[f"{n}.0.bias"] = (256,)
        s[f"{n}.2.weight"] = (d_z, 256); s[f"{n}.2.bias"] = (d_z,)
    for l in range(n_dec):
        s[f"decoder_rnn.weight_ih_l{l}"] = (3 * d_h, V + d_z if l == 0 else d_h)
        s[f"decoder_rnn.weight_hh_l{l}"] = (3 * d_h, d_h)
        s[f"decoder_rnn.bias_ih_l{l}"] = (3 * d_h,); s[f"decoder_rnn.bias_hh_l{l}"] = (3 * d_h,)
    s["decoder_lat.weight"] = (d_h, d_z); s["decoder_lat.bias"] = (d_h,)
    s["decoder_fc.weight"] = (V, d_h); s["decoder_fc.bias"] = (V,)
    return s


def models2d_shapes(H=501, C=35, L=120):
    """models2d.VAE state_dict (models2d.py:12-21)."""
    s = {"conv1d1.weight": (9, L, 9), "conv1d1.bias": (9,), "conv1d2.weight": (9, 9, 9), "conv1d2.bias": (9,),
         "conv1d3.weight": (10, 9, 11), "conv1d3.bias": (10,), "fc0.weight": (435, 90), "fc0.bias": (435,),
         "fc11.weight": (2, 435), "fc11.bias": (2,), "fc12.weight": (2, 435), "fc12.bias": (2,), "fc2.weight": (2, 2), "fc2.bias": (2,)}
    for l in range(3):
        s[f"gru.weight_ih_l{l}"] = (3 * H, 2 if l == 0 else H); s[f"gru.weight_hh_l{l}"] = (3 * H, H)
        s[f"gru.bias_ih_l{l}"] = (3 * H,); s[f"gru.bias_hh_l{l}"] = (3 * H,)
    s["fc3.weight"] = (C, H); s["fc3.bias"] = (C,)
    return s


def init_params(shapes, seed, gain=1.0, dtype=np.float32):
    """uniform(-a, a), a = gain / sqrt(fan_in) (biases: a = gain * 0.1); one independent
    stream per key (seed mixed with crc32 of the key) so subsets reproduce."""
    out = {}
    for k in sorted(shapes):
        shp = shapes[k]
        rs = np.random.RandomState((seed * 1000003 + zlib.crc32(k.encode())) % (2 ** 31 - 1))
        if len(shp) == 1:
            a = 0.1 * gain
        else:
            a = gain / np.sqrt(float(np.prod(shp[1:])))
        out[k] = rs.uniform(-a, a, size=shp).astype(dtype)
    return out


def seeded_indices(seed, B, L, C):
    return np.random.RandomState(seed).randint(0, C, size=(B, L)).astype(np.int64)


def seeded_eps(seed, B, o, scale=1e-2, dtype=np.float32):
    return (scale * np.random.RandomState(seed + 7919).standard_normal((B, o))).astype(dtype)
