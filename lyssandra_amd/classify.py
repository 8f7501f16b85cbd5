"""Host-side evaluation harness of the classifiers (lyssa/classify.py:7-187) -- control flow only, no device work.

Kept because `lc_ksvd_classifier` (config 5) is driven through it: `clf(X, y)` splits the dataset (stratified folds, or
`n_class_samples` training / `n_test_samples` test columns per class drawn with the GLOBAL numpy RNG in the reference's
order, utils/dataset.py:241-265), trains for every parameter set of `param_grid` and keeps the best one.
"""
import itertools

import numpy as np


def class_accuracy(y_pred, y_test):
    """lyssa/classify.py:9-12."""
    return np.sum(np.asarray(y_test) == np.asarray(y_pred)) / float(np.asarray(y_test).size)


def avg_class_accuracy(y_pred, y_test):
    """lyssa/classify.py:15-25: accuracy averaged over the classes present in y_test."""
    y_pred, y_test = np.asarray(y_pred), np.asarray(y_test)
    per_class = [np.mean(y_pred[y_test == c] == c) for c in range(len(set(y_test.tolist())))]
    return float(np.mean(per_class))


def split_dataset(n_training_samples, n_test_samples, y):
    """lyssa/utils/dataset.py:241-265 (labels sorted by class: class c occupies one contiguous range of columns).
    Per class: one `np.random.choice` for the training columns, a second one only when the remaining columns exceed
    `n_test_samples[c]`."""
    y = np.asarray(y)
    train, test, offset = [], [], 0
    for c in range(len(set(y.tolist()))):
        size = int(np.sum(y == c))
        tr = np.random.choice(size, size=n_training_samples[c], replace=False)
        te = np.setdiff1d(np.arange(size), tr)
        if n_test_samples is not None and te.size > n_test_samples[c]:
            te = te[np.random.choice(te.size, size=n_test_samples[c], replace=False)]
        train.append(offset + tr)
        test.append(offset + te)
        offset += size
    return [np.concatenate(train).astype(int), np.concatenate(test).astype(int)]


def parameter_grid(param_grid):
    """The parameter sets of a list of {name: [values]} dicts, in sklearn `ParameterGrid` order (names sorted, last
    name varies fastest)."""
    grids = [param_grid] if isinstance(param_grid, dict) else list(param_grid)
    for grid in grids:
        names = sorted(grid)
        for values in itertools.product(*(grid[name] for name in names)):
            yield dict(zip(names, values))


class classifier(object):
    """lyssa/classify.py:83-187: `train` / `predict` are supplied by the subclass."""

    def __init__(self, param_grid=None, n_folds=None, n_class_samples=None, n_test_samples=None, n_tests=1,
                 name="classifier"):
        self.name = name
        self.param_grid = param_grid
        self.best_param_set = None
        self.best_score = None
        self.n_folds = n_folds
        self.n_test_samples = n_test_samples      # validation / test samples per class
        self.n_class_samples = n_class_samples    # training samples per class
        self.n_tests = n_tests
        self.folds = []

    def fit(self, X, y):
        self.__call__(X, y)

    def __call__(self, X, y):
        y = np.asarray(y)
        n_classes = len(set(y.tolist()))
        if self.n_folds is not None:
            from sklearn.model_selection import StratifiedKFold
            self.folds = list(StratifiedKFold(n_splits=self.n_folds, shuffle=False).split(np.zeros(y.size), y))
        elif self.n_class_samples is not None:
            if not isinstance(self.n_class_samples, (list, np.ndarray)):
                self.n_class_samples = (np.ones(n_classes) * self.n_class_samples).astype(int)
            if self.n_test_samples is not None and not isinstance(self.n_test_samples, (list, np.ndarray)):
                self.n_test_samples = (np.ones(n_classes) * self.n_test_samples).astype(int)
            self.folds = [tuple(split_dataset(self.n_class_samples, self.n_test_samples, y)) for _ in range(self.n_tests)]
        self.cross_validate(X, y)

    def cross_validate(self, X, y):
        if self.param_grid is None:
            self.best_score = self.evaluate(X, y)
            return
        scores, sets = [], list(parameter_grid(self.param_grid))
        for param_set in sets:
            scores.append(self.evaluate(X, y, param_set=param_set))
        best = int(np.argmax(np.array(scores)))
        self.best_param_set, self.best_score = sets[best], scores[best]

    def evaluate(self, X, y, param_set=None):
        """Mean accuracy over the folds of the classifier trained with `param_set`."""
        scores = []
        for train_index, test_index in self.folds:
            self.train(X[:, train_index], y[train_index], param_set=param_set)
            scores.append(class_accuracy(np.array(self.predict(X[:, test_index])), y[test_index]))
        return float(np.mean(scores))

    def train(self, X_train, y_train, param_set=None):
        raise NotImplementedError

    def predict(self, X_test):
        raise NotImplementedError
