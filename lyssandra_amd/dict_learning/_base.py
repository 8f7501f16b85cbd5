"""Shared shell of the drop-in learner classes.

The reference's `ksvd_coder`, `online_dictionary_coder` and `dictionary_learner` (lyssa/dict_learning/ksvd.py:234-271,
online_dict_learn.py:127-160, gradient_descent.py:128-160) are three keyword-argument holders around a `*_dict_learn`
function with the same three public methods.  Their constructor signatures are API (kept verbatim in the subclasses);
everything else lives here once: the constructor arguments are stored under their own names, `fit` forwards the
subset its learner accepts, `encode` / `__call__` run the sparse coder against the learned dictionary."""


class learner_shell(object):
    _forward = ()          # constructor arguments handed to the learner function by `fit`
    _rename = {}           # constructor name -> learner keyword, where they differ

    def _hold(self, ctor_locals):
        for name, value in ctor_locals.items():
            if name != "self" and not name.startswith("__"):
                setattr(self, name, value)
        self.D = None

    def _learner_kwargs(self):
        return {self._rename.get(name, name): getattr(self, name) for name in self._forward}

    def _learn(self, X):   # subclasses: run the learner, set self.D (and whatever else they keep)
        raise NotImplementedError

    def fit(self, X):
        self._learn(X)

    def encode(self, X):
        return self.sparse_coder(X, self.D)

    def __call__(self, X):
        self._learn(X)
        return self.encode(X)


class reference_patience(object):
    """The stopping bookkeeping the reference's three learners share (ksvd.py:222-229, online_dict_learn.py:112-118,
    gradient_descent.py:112-122), reproduced ONCE with its quirk: the previous error is refreshed only when `verbose`,
    and BEFORE the comparison, so from the second evaluation on every evaluation counts as "no improvement" (with
    verbose the error is compared with itself, without it with 0) and the learners stop after `limit` + 1 of them."""
    limit = 10

    def __init__(self, verbose):
        self.verbose = bool(verbose)
        self.previous = 0
        self.strikes = 0

    def change(self, error):
        """error - previous error, for the reference's progress line (call before `observe`)."""
        return error - self.previous

    def observe(self, step, error):
        if self.verbose:
            self.previous = error
        if step > 0 and (error > 0.9 * self.previous or error > self.previous):
            self.strikes += 1

    @property
    def exhausted(self):
        return self.strikes >= self.limit


def starting_dictionary(X, n_atoms, D_init):
    """`D_init` itself (the learners update it in place, like the reference) or `init_dictionary(method='data')`."""
    if D_init is not None:
        return D_init
    from .utils import init_dictionary
    return init_dictionary(X, n_atoms, method='data', return_unused_data=True)[0]
