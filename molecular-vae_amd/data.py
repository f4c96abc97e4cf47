"""Input side of the training path: the ``(LongTensor[L], FloatTensor[L, C])`` contract of ``data_loader.MoleLoader``
(data_loader.py:26-31) without its per-item sklearn ``OneHotEncoder.fit_transform``, plus a device-resident variant.

  * ``build_vocab``     -- train.py:45-60 with a *sorted* character order (the reference iterates a ``set``, so its ids are not
                           reproducible across runs); ``' '`` is the padding character (``ljust``), as in the reference.
  * ``MoleLoader``      -- same constructor / ``__getitem__`` contract as the reference class (drop-in for ``DataLoader``).
  * ``encode_smiles``   -- whole-corpus vectorised encoding to uint8 ``[N, L]`` (120 B / molecule; 250k ZINC = 30 MB).
  * ``DeviceDataset``   -- the encoded corpus resident in HBM; per-epoch shuffled, rank-sharded batches expanded on device
                           (``mvae_expand_indices``) to the same ``(idx, ohe)`` pair -- no host work in the step.
  * ``load_smiles`` / ``save_encoded`` / ``load_encoded`` -- ``.smi``/CSV in, ``.npz`` (indices + charset + max_len) out.
"""
import numpy as np
import torch


def build_vocab(smiles, max_len=None):
    """char -> id over every string shorter than max_len (train.py:47-54), plus ' '; sorted for reproducibility."""
    chars = set(" ")
    for s in smiles:
        if max_len is None or len(s) < max_len:
            chars.update(s)
    return {ch: i for i, ch in enumerate(sorted(chars))}


def encode_smiles(smiles, vocab, max_len):
    """uint8 [N, max_len]; strings are right-padded with ' ' (str.ljust, data_loader.py:27); unknown chars raise KeyError
    (as the reference's dict lookup does); longer strings raise ValueError."""
    if len(vocab) > 256:
        raise ValueError("vocabulary does not fit uint8")
    out = np.full((len(smiles), max_len), vocab[" "], dtype=np.uint8)
    for n, s in enumerate(smiles):
        if len(s) > max_len:
            raise ValueError(f"SMILES longer than max_len={max_len}: {s!r}")
        out[n, :len(s)] = [vocab[ch] for ch in s]
    return out


class MoleLoader(torch.utils.data.Dataset):
    """data_loader.py:7-31: ``df`` is anything indexable whose rows' first column is the SMILES string (a pandas DataFrame, as
    in the reference, or a plain list of strings)."""

    def __init__(self, df, vocab, max_len=70, num=None):
        super().__init__()
        self.df, self.vocab, self.max_len = df, vocab, max_len
        self._eye = np.eye(len(vocab), dtype=np.float32)

    def __len__(self):
        return self.df.shape[0] if hasattr(self.df, "shape") else len(self.df)

    def _smile(self, item):
        return str(self.df.iloc[item, 0]) if hasattr(self.df, "iloc") else str(self.df[item])

    def __getitem__(self, item):
        smile = self._smile(item).ljust(self.max_len, " ")
        embedding = np.array([self.vocab[ch] for ch in smile])
        return torch.LongTensor(embedding), torch.from_numpy(self._eye[embedding])


class DeviceDataset:
    """uint8 indices [N, L] in HBM; ``batches()`` yields (LongTensor[B,L], FloatTensor[B,L,C]) device tensors."""

    def __init__(self, indices_u8, n_classes, device="cuda"):
        self.store = torch.as_tensor(np.ascontiguousarray(indices_u8), dtype=torch.uint8).to(device)
        self.n, self.L = self.store.shape
        self.C = int(n_classes)
        self.device = self.store.device

    def __len__(self):
        return self.n

    def epoch_order(self, epoch=0, seed=0, shuffle=True, rank=0, world=1):
        """This rank's contiguous shard of the (optionally shuffled) epoch permutation, on device."""
        if shuffle:
            g = torch.Generator(); g.manual_seed(seed + epoch)
            perm = torch.randperm(self.n, generator=g)
        else:
            perm = torch.arange(self.n)
        per = self.n // world
        return perm[rank * per:(rank + 1) * per].to(self.device)

    def batches(self, batch_size, epoch=0, seed=0, shuffle=True, rank=0, world=1, drop_last=True, want_onehot=True):
        from . import ops
        order = self.epoch_order(epoch, seed, shuffle, rank, world)
        n = order.numel()
        stop = n - n % batch_size if drop_last else n
        for lo in range(0, stop, batch_size):
            rows = order[lo:lo + batch_size].contiguous()
            B = rows.numel()
            idx = torch.empty(B, self.L, dtype=torch.long, device=self.device)
            ohe = torch.empty(B, self.L, self.C, dtype=torch.float32, device=self.device) if want_onehot else None
            ops.expand_indices(self.store, rows, idx, ohe, B, self.L, self.C)
            yield idx, ohe


def synthetic_smiles(n, seed=0, lo=20, hi=60, structured=True):
    """A stand-in corpus over the ZINC alphabet (no data set ships with the reference: .MISSING_LARGE_BLOBS).  structured: every string is
    a short random motif repeated to its length, so that a model can learn something in a few hundred steps (tests / examples);
    otherwise i.i.d. characters."""
    rs = np.random.RandomState(seed)
    alphabet = list("CNOSFcnos()=#123[]@H+-lBr")
    out = []
    for _ in range(n):
        ln = int(rs.randint(lo, hi))
        if structured:
            m = rs.choice(alphabet, size=int(rs.randint(3, 8)))
            out.append("".join(np.resize(m, ln)))
        else:
            out.append("".join(rs.choice(alphabet, size=ln)))
    return out


def indices_to_smiles(idx, charset):
    """Rows of class indices -> strings, right-stripped of the padding character (train_sample.py:36: "".join(charset[i]).rstrip()).
    `charset`: id -> char (dict or list), the ``charset`` entry of the reference's checkpoints (train.py:174)."""
    arr = np.asarray(idx.detach().cpu() if hasattr(idx, "detach") else idx)
    if isinstance(charset, dict):
        table = np.array([charset[i] for i in range(len(charset))])
    else:
        table = np.array(list(charset))
    return ["".join(table[row]).rstrip() for row in arr]


def load_smiles(path, column=0):
    """.smi / headerless CSV: one molecule per line, SMILES in `column` (train.py:41 pd.read_csv(..., header=None))."""
    out = []
    with open(path) as f:
        for line in f:
            line = line.strip()
            if line:
                out.append(line.replace("\t", ",").split(",")[column].split()[0])
    return out


def save_encoded(path, indices_u8, vocab, max_len):
    chars = sorted(vocab, key=vocab.get)
    np.savez_compressed(path, indices=indices_u8, charset=np.array(chars), max_len=np.int64(max_len))


def load_encoded(path):
    z = np.load(path)
    chars = [str(c) for c in z["charset"]]
    return z["indices"], {ch: i for i, ch in enumerate(chars)}, int(z["max_len"])
