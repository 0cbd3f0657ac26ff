"""Score -> probability calibrators (seesaw/calibration.py:28-57).  Host side, tiny."""
from __future__ import annotations

import numpy as np
import scipy.special


class FixedCalibrator:
    def __init__(self, a: float, b: float, sigmoid: bool):
        self.a, self.b, self.sigmoid = a, b, sigmoid

    def get_probabilities(self, vector_scorer, vectors):
        rescaled = self.a * (vectors @ vector_scorer.reshape(-1) + self.b)
        return scipy.special.expit(rescaled) if self.sigmoid else rescaled


class GroundTruthCalibrator:
    """Platt scaling fitted on ground truth (debug / experiments only, calibration.py:28-42)."""

    def __init__(self, X, y):
        assert X.shape[0] == y.shape[0]
        self.X, self.y = X, y
        self._mean = y.mean()

    def get_mean(self):
        return self._mean

    def get_probabilities(self, vector_scorer, vectors):
        from sklearn.calibration import _SigmoidCalibration
        sc = _SigmoidCalibration()
        sc.fit((self.X @ vector_scorer.reshape(-1)).reshape(-1, 1), self.y)
        return sc.predict(vectors @ vector_scorer.reshape(-1))
