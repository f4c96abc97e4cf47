"""Input-pipeline contract (data_loader.py:26-31) on CPU, and the device-side expansion on the GPU."""
import os

import numpy as np
import pytest
import torch

import molecular_vae_amd as mv
from molecular_vae_amd import data as D

SMILES = ["CCO", "c1ccccc1", "CC(=O)Oc1ccccc1C(=O)O", "CN1CCC[C@H]1c2cccnc2", "O=C(O)c1ccccc1", "C1CC1", "N#Cc1ccc(Br)cc1"]


def test_vocab_and_moleloader_contract():
    vocab = D.build_vocab(SMILES, max_len=40)
    assert vocab[" "] == 0 and list(vocab) == sorted(vocab)             # deterministic ids, ' ' is the ljust pad
    ds = D.MoleLoader(SMILES, vocab, max_len=40)
    idx, ohe = ds[2]
    assert idx.dtype == torch.long and idx.shape == (40,) and ohe.dtype == torch.float32 and ohe.shape == (40, len(vocab))
    assert "".join(sorted(vocab, key=vocab.get)[i] for i in idx.tolist()).rstrip() == SMILES[2]
    assert torch.equal(ohe.argmax(1), idx) and float(ohe.sum()) == 40.0
    loader = torch.utils.data.DataLoader(ds, batch_size=3)
    a, b = next(iter(loader))
    assert a.shape == (3, 40) and b.shape == (3, 40, len(vocab))
    with pytest.raises(KeyError):
        D.MoleLoader(["C?"], vocab, 40)[0]                             # unknown character: KeyError, as the reference's dict lookup


def test_encode_roundtrip_and_npz(tmp_path):
    vocab = D.build_vocab(SMILES)
    enc = D.encode_smiles(SMILES, vocab, 32)
    assert enc.dtype == np.uint8 and enc.shape == (len(SMILES), 32)
    ds = D.MoleLoader(SMILES, vocab, 32)
    for n in range(len(SMILES)):
        assert (enc[n] == ds[n][0].numpy()).all()
    p = os.path.join(str(tmp_path), "enc.npz")
    D.save_encoded(p, enc, vocab, 32)
    enc2, vocab2, L = D.load_encoded(p)
    assert (enc2 == enc).all() and vocab2 == vocab and L == 32
    smi = os.path.join(str(tmp_path), "x.smi")
    open(smi, "w").write("\n".join(SMILES) + "\n")
    assert D.load_smiles(smi) == SMILES
    with pytest.raises(ValueError):
        D.encode_smiles(["C" * 40], vocab, 32)


@pytest.mark.gpu
def test_device_dataset_expansion_matches_moleloader():
    vocab = D.build_vocab(SMILES)
    enc = D.encode_smiles(SMILES * 10, vocab, 32)
    dd = D.DeviceDataset(enc, len(vocab))
    seen = 0
    ref = D.MoleLoader(SMILES * 10, vocab, 32)
    for rank in range(2):
        order = dd.epoch_order(epoch=3, seed=1, rank=rank, world=2).cpu()
        for k, (idx, ohe) in enumerate(dd.batches(8, epoch=3, seed=1, rank=rank, world=2)):
            rows = order[k * 8:(k + 1) * 8]
            for j, r in enumerate(rows.tolist()):
                ri, ro = ref[r]
                assert torch.equal(idx[j].cpu(), ri) and torch.equal(ohe[j].cpu(), ro)
            seen += idx.shape[0]
    assert seen == 64
