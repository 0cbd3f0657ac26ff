"""Score -> probability calibrators (seesaw/calibration.py:28-57).  Host side, tiny."""
from __future__ import annotations

import numpy as np
import scipy.special


class FixedCalibrator:
    def __init__(self, a: float, b: float, sigmoid: bool):
        self.a, self.b, self.sigmoid = a, b, sigmoid

    def get_probabilities(self, vector_scorer, vectors, scores=None):
        """`scores`: the index's own scan of `vector_scorer` (AccessMethod.score, on the device) when the caller
        already has it -- the reference's `vectors @ vector_scorer` (calibration.py:51) over the whole index is the
        scan this package exists to replace; without it the expression runs as the reference writes it."""
        if scores is None:
            scores = vectors @ vector_scorer.reshape(-1)
        rescaled = self.a * (np.asarray(scores) + self.b)
        return scipy.special.expit(rescaled) if self.sigmoid else rescaled


class GroundTruthCalibrator:
    """Platt scaling fitted on ground truth (debug / experiments only, calibration.py:28-42)."""

    def __init__(self, X, y):
        assert X.shape[0] == y.shape[0]
        self.X, self.y = X, y
        self._mean = y.mean()

    def get_mean(self):
        return self._mean

    def get_probabilities(self, vector_scorer, vectors, scores=None):
        from sklearn.calibration import _SigmoidCalibration
        sc = _SigmoidCalibration()
        sc.fit((self.X @ vector_scorer.reshape(-1)).reshape(-1, 1), self.y)
        return sc.predict(vectors @ vector_scorer.reshape(-1) if scores is None else np.asarray(scores))
