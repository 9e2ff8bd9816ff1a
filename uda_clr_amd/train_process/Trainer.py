"""Placeholder for the reference's BEAL trainer (``train_process/Trainer.py``): it expects a 3-tuple
generator and is unreachable from ``train_use_fix_initial.py:258-304`` (out of
scope).  The module exists because the entry script imports it by name."""


class Trainer(object):
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("the BEAL Trainer is outside the built hot path; use "
                                  "Trainer_baseline.Trainer or Trainer_prototype_full.Trainer")
