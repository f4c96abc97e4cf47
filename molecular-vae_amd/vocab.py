"""Character vocabulary of the MOSES path (behaviour of vocab.py:10-87 / mosesvocab.py:26-103).

Symbols are the sorted characters of the corpus followed by the four specials <bos>, <eos>, <pad>, <unk> (in that
order), so ids are reproducible.  ``OneHotVocab.vectors`` is the identity that initialises ``VAE.x_emb`` (mosesvae.py:48-50).
"""
import torch


class SS:
    bos, eos, pad, unk = "<bos>", "<eos>", "<pad>", "<unk>"


class CharVocab:
    def __init__(self, chars, ss=SS):
        specials = [ss.bos, ss.eos, ss.pad, ss.unk]
        if any(sp in chars for sp in specials):
            raise ValueError("SS in chars")
        self.ss = ss
        symbols = sorted(chars) + specials
        self.c2i = dict(zip(symbols, range(len(symbols))))
        self.i2c = dict(enumerate(symbols))

    @classmethod
    def from_data(cls, data, *args, **kwargs):
        return cls(set().union(*map(set, data)) if data else set(), *args, **kwargs)

    def __len__(self):
        return len(self.c2i)

    bos = property(lambda self: self.c2i[self.ss.bos])
    eos = property(lambda self: self.c2i[self.ss.eos])
    pad = property(lambda self: self.c2i[self.ss.pad])
    unk = property(lambda self: self.c2i[self.ss.unk])

    def char2id(self, char):
        return self.c2i.get(char, self.unk)

    def id2char(self, id):
        return self.i2c.get(id, self.ss.unk)

    def string2ids(self, string, add_bos=False, add_eos=False):
        ids = [self.char2id(ch) for ch in string]
        return ([self.bos] if add_bos else []) + ids + ([self.eos] if add_eos else [])

    def ids2string(self, ids, rem_bos=True, rem_eos=True):
        ids = list(ids)
        if ids and rem_bos and ids[0] == self.bos:
            ids = ids[1:]
        if ids and rem_eos and ids[-1] == self.eos:
            ids = ids[:-1]
        return "".join(self.id2char(i) for i in ids)


class OneHotVocab(CharVocab):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.vectors = torch.eye(len(self.c2i))


def string2tensor(vocab, string):
    """moses_train_distrib.py:91-97: <bos> + ids + <eos>, int64."""
    return torch.tensor(vocab.string2ids(string, add_bos=True, add_eos=True), dtype=torch.long)


def get_collate_fn(vocab):
    """moses_train_distrib.py:127-135: sort the strings by length, longest first (stable), then tokenise."""
    def collate(data):
        data = sorted(data, key=len, reverse=True)
        return [string2tensor(vocab, s) for s in data]
    return collate


class PaddedBatch:
    """A collated batch already in the layout the kernels read: ``x_pad`` int64 [B, T] (pad-filled, rows sorted by length descending)
    and ``lengths`` int32 [B], both on the target device.  ``mosesvae.VAE.forward`` takes it in place of the list of per-sequence
    tensors: ONE host->device transfer per batch instead of the reference's one ``.cuda()`` per sequence (moses_train_distrib.py:271)."""

    def __init__(self, x_pad, lengths):
        self.x_pad, self.lengths = x_pad, lengths

    def __len__(self):
        return self.x_pad.shape[0]

    def to(self, device, non_blocking=True):
        return PaddedBatch(self.x_pad.to(device, non_blocking=non_blocking), self.lengths.to(device, non_blocking=non_blocking))

    def tensors(self):
        """The reference's representation (list of LongTensors) -- for code that still wants it."""
        return [self.x_pad[b, :int(n)] for b, n in enumerate(self.lengths.tolist())]


def pad_batch(tensors, pad):
    """list of LongTensors sorted by length descending -> PaddedBatch (on the tensors' device)."""
    lengths = [int(t.numel()) for t in tensors]
    if any(lengths[i] < lengths[i + 1] for i in range(len(lengths) - 1)):
        raise RuntimeError("sequences must be sorted by length in decreasing order (pack_sequence, mosesvae.py:151)")
    x_pad = torch.nn.utils.rnn.pad_sequence(list(tensors), batch_first=True, padding_value=pad)
    return PaddedBatch(x_pad, torch.tensor(lengths, dtype=torch.int32, device=x_pad.device))


def get_padded_collate_fn(vocab, pin_memory=False):
    """Like ``get_collate_fn`` but returns a PaddedBatch (host tensors, optionally pinned) ready for one asynchronous ``.to(device)``."""
    inner = get_collate_fn(vocab)

    def collate(data):
        b = pad_batch(inner(data), vocab.pad)
        if pin_memory:
            b = PaddedBatch(b.x_pad.pin_memory(), b.lengths.pin_memory())
        return b
    return collate
