"""MI355X-native SMILES-VAE training hot path (drop-in for aclyde11/molecular-VAE's models.py / train.py surface).

The directory name (``molecular-vae_amd``) is fixed by the repo layout contract and is not a Python identifier;
import it through the ``molecular_vae_amd`` alias package at the repo root.
"""
from . import _lib  # noqa: F401
from .models import (MolecularVAE, MolEncoder, MolDecoder, Lambda, ConvSELU, SELU, TimeDistributed, Repeat,  # noqa: F401
                     Flatten)
from .functional import bce_kl_loss, make_loss_function  # noqa: F401
from . import mosesvae, models2d, vocab, data  # noqa: F401
from .data import MoleLoader, DeviceDataset, build_vocab, encode_smiles, synthetic_smiles, indices_to_smiles  # noqa: F401
from .vocab import CharVocab, OneHotVocab, PaddedBatch, pad_batch, get_collate_fn, get_padded_collate_fn  # noqa: F401
from .train import (FusedAdam, GradSync, ShardedSampler, shard_batch, train_step, exact_match_accuracy, evaluate, save_checkpoint,  # noqa: F401
                    load_checkpoint, strip_module_prefix, KLAnnealer, CosineAnnealingLRWithRestart, cosine_lr_with_restart,
                    moses_train_step, moses_train_epoch, generate_from_latent)

__all__ = ["mosesvae", "models2d", "vocab", "data", "MoleLoader", "DeviceDataset", "build_vocab", "encode_smiles", "CharVocab", "OneHotVocab", "MolecularVAE", "MolEncoder", "MolDecoder", "Lambda", "ConvSELU", "SELU", "TimeDistributed", "Repeat",
           "Flatten", "bce_kl_loss", "make_loss_function", "FusedAdam", "GradSync", "ShardedSampler", "shard_batch",
           "train_step", "exact_match_accuracy", "evaluate", "save_checkpoint", "load_checkpoint", "strip_module_prefix", "KLAnnealer",
           "CosineAnnealingLRWithRestart", "cosine_lr_with_restart", "moses_train_step", "moses_train_epoch", "generate_from_latent", "synthetic_smiles", "indices_to_smiles"]
