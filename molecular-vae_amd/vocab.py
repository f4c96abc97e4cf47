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
