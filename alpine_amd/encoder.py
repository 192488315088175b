"""One-hot encoding of covariate columns -- host-side ingest of the fit path.

Same behaviour as ``FeatureEncoders`` (alpine/utils/encoder.py:11-60), which wraps
``sklearn.preprocessing.OneHotEncoder(sparse_output=False, handle_unknown="ignore")``:
categories are the sorted distinct non-missing values seen at fit time, missing values
(NaN / None) become all-zero rows, values unseen at fit time become all-zero rows at transform
time, and the column names are ``"<key>_<label>"``.  Restated with numpy/pandas only."""
from __future__ import annotations

from typing import Dict, List

import numpy as np
import pandas as pd


class FeatureEncoders:
    def __init__(self, covariate_keys: List[str]):
        self.covariate_keys: List[str] = covariate_keys
        self.categories: Dict[str, np.ndarray] = {}
        self.encoded_labels: Dict[str, List[str]] = {}

    def _encode(self, col: pd.Series, cats: np.ndarray) -> np.ndarray:
        na = col.isna().to_numpy()
        out = np.zeros((len(col), len(cats)), dtype=np.float32)
        vals = col.to_numpy()[~na]
        idx = np.searchsorted(cats, vals)
        idx_c = np.clip(idx, 0, len(cats) - 1)
        known = cats[idx_c] == vals
        rows = np.flatnonzero(~na)[known]
        out[rows, idx_c[known]] = 1.0
        return out

    def fit_transform(self, df: pd.DataFrame, merge_categories=None) -> List[np.ndarray]:
        """``merge_categories`` (extension for rank-local inputs): a callable that maps this rank's sorted categories
        of one column to the sorted union over all ranks, so that every rank builds the same one-hot columns."""
        if not isinstance(df, pd.DataFrame):
            raise TypeError("adata.obs must be a pandas DataFrame.")
        mats = []
        for key in self.covariate_keys:
            col = df[key]
            cats = np.unique(col[~col.isna()].to_numpy())          # sorted, like OneHotEncoder(categories="auto")
            if merge_categories is not None:
                cats = merge_categories(cats)
            self.categories[key] = cats
            self.encoded_labels[key] = [f"{key}_{c}" for c in cats.tolist()]
            mats.append(self._encode(col, cats))
        return mats

    def transform(self, df: pd.DataFrame) -> List[np.ndarray]:
        if not isinstance(df, pd.DataFrame):
            raise TypeError("adata.obs must be a pandas DataFrame.")
        return [self._encode(df[key], self.categories[key]) for key in self.covariate_keys if key in self.categories]
