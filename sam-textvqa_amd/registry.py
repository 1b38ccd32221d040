"""Process-global attribute dict mirroring /root/reference/tools/registry.py:1-3 (an EasyDict there).
SAM4C reads `registry.answer_vocab` (its length sizes the classifier, sa_m4c.py:169) and `registry.BOS_IDX` (:291)."""


class _Registry(dict):
    __getattr__ = dict.get

    def __setattr__(self, k, v):
        self[k] = v


registry = _Registry()
