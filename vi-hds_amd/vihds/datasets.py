"""Plate-reader CSV loading, multi-file merge, scaling, device one-hots and the cross-validation split
(counterpart of the reference's vihds/datasets.py + data/procdata.py).  Host I/O only -- out of the hot path, kept
so that `run_xval.py <spec>.yaml` works on the reference's data files unchanged."""
import os
from collections import OrderedDict

import numpy as np
import pandas as pd
import torch
from torch.utils.data import Dataset, Subset


def onehot(i, n):
    v = np.zeros((n))
    if i is not None:
        v[i] = 1
    return v


def depth(group_values):
    return len(set(g for g in group_values if g is not None))


def get_cassettes(devices, settings):
    """Concatenated one-hot blocks, one block per component group (reference datasets.py:26-45)."""
    rows = []
    for d in devices:
        name = settings.device_idx_to_device_name[d]
        rows.append(np.hstack([onehot(cm[name], depth(cm.values())) for cm in settings.component_maps.values()]))
    return np.array(rows).astype(np.float32)


def scale_data(X, settings):
    """Per-signal scaling to max 1 (or `normalize`), then per-series background subtraction (datasets.py:48-61)."""
    n_outputs = np.shape(X)[1]
    scales = settings.normalize if settings.normalize is not None else [
        np.max(X[:, i, :]).astype(np.float32) for i in range(n_outputs)]
    for i, scale in enumerate(scales):
        X[:, i, :] /= scale
        if settings.subtract_background:
            X[:, i, :] -= np.min(X[:, i, :], axis=1)[:, np.newaxis]
    return X, scales


def _parse_condition(text):
    """'C6=25000;C12=5' -> OrderedDict (reference data/procdata.py:15-28)."""
    d = OrderedDict()
    if "=" in text:
        for item in text.split(";"):
            k, v = item.split("=")
            d[k] = float(v)
    return d


def _signal_of(header):
    """'Raw Data (EYFP) 3 - 0 h 23 min' -> 'EYFP' (reference data/procdata.py:63-73)."""
    a = header.find("(")
    if a >= 0:
        b = header.find(")")
        if b >= 0:
            return header[a + 1: b]
    return header


def load_csv(csv_file, settings):
    """One plate-reader file -> (devices [L], treatments [L,C], times [T], observations [L,4,T])
    (reference data/procdata.py:122-187): row 0 holds the time of every reading column; rows are kept when their
    device is listed in the spec and every condition not listed in the spec is zero."""
    table = pd.read_csv(os.path.join(settings.data_dir, csv_file), sep=",", na_filter=False)
    timesall = table.iloc[0, 5:]
    body = table.iloc[1:, :]
    body = body.iloc[np.isin(body.iloc[:, 0], settings.devices), :]
    devices = np.array([settings.device_map[dev] for dev in body.iloc[:, 0]], dtype=int)
    treatments = [_parse_condition(c) for c in body.iloc[:, 4]]
    if len(treatments) == 0:
        return None
    keep, kept = [], []
    for i, tr in enumerate(treatments):
        if all(v == 0.0 for k, v in tr.items() if k not in settings.conditions):
            keep.append(i)
            kept.append([tr.get(c, 0.0) for c in settings.conditions])
    readings = body.iloc[keep, 5:]
    signals = np.array([_signal_of(str(h).split(".")[0]) for h in readings.columns.values])
    obs = np.array([[row.iloc[signals == s].values for s in settings.signals] for _, row in readings.iterrows()])
    times = timesall.iloc[signals == "OD"].values
    return (devices[keep] if len(keep) != len(devices) else devices, np.array(kept).astype(np.float32),
            times.astype(np.float32), obs.astype(np.float32))


def find_nearest(array, value):
    return (np.abs(np.asarray(array) - value)).argmin()


def merge_observations(times_list, observations_list):
    """Resample every file onto the shortest file's time grid by nearest time (reference datasets.py:136-145;
    ragged inputs are kept as lists -- the reference's np.asarray of a ragged list fails on numpy >= 1.24)."""
    loc = int(np.argmin([len(t) for t in times_list]))
    chosen = times_list[loc]
    out = []
    for t, obs in zip(times_list, observations_list):
        locs = [find_nearest(t, ti) for ti in chosen]
        out.append(obs[:, :, locs])
    return chosen, np.concatenate(out)


class TimeSeriesDataset(Dataset):
    """reference datasets.py:64-124"""

    def __init__(self, data_settings, params):
        self.data_settings = data_settings
        self.params = params
        self.n_times = None
        self.n_species = None

    def _preprocess(self, devices, inputs, times, observations):
        self.devices = devices
        self.dev_1hot = torch.tensor(get_cassettes(devices, self.data_settings))
        self.inputs = torch.tensor(np.log(1.0 + inputs))
        self.times = torch.tensor(times)
        self.n_times = len(times)
        obs, self.scales = scale_data(observations, self.data_settings)
        self.observations = torch.tensor(obs)
        self.n_species = np.shape(observations)[1]

    def init_single(self, f):
        self._preprocess(*load_csv(f, self.data_settings))

    def init_multiple_merge(self):
        loaded = [load_csv(f, self.data_settings) for f in self.data_settings.files]
        devices, inputs, times_list, obs_list = zip(*loaded)
        times, observations = merge_observations(list(times_list), list(obs_list))
        self._preprocess(np.concatenate(devices), np.concatenate(inputs), times, observations)

    def __len__(self):
        return len(self.devices)

    def __getitem__(self, idx):
        if torch.is_tensor(idx):
            idx = idx.tolist()
        return {"devices": self.devices[idx], "dev_1hot": self.dev_1hot[idx], "inputs": self.inputs[idx],
                "observations": self.observations[idx]}


class TimeSeriesDatasetPair(object):
    """reference datasets.py:148-170"""

    def __init__(self, train_dataset, test_dataset, data_settings):
        self.train, self.test = train_dataset, test_dataset
        self.n_train, self.n_test = len(train_dataset), len(test_dataset)
        self.depth = data_settings.device_depth
        self.n_conditions = len(data_settings.conditions)


def split_dataset(dataset, args, data_settings):
    """Seeded permutation split into `folds` chunks; chunk `split` is the validation set (datasets.py:199-222)."""
    np.random.seed(args.seed)
    if getattr(args, "heldout", None):
        raise NotImplementedError("TODO: implement heldout device")
    n = len(dataset)
    val_chunks = np.array_split(np.random.permutation(n), args.folds)
    val_ids = np.sort(val_chunks[args.split - 1])
    train_ids = np.setdiff1d(np.arange(n, dtype=int), val_ids)
    return TimeSeriesDatasetPair(Subset(dataset, train_ids), Subset(dataset, val_ids), data_settings)


def build_datasets(args, config):
    """reference datasets.py:173-224"""
    if not config.data.merge:
        raise NotImplementedError("Can't handle multiple datasets yet")  # as the reference's Encoder (encoders.py:363)
    dataset = TimeSeriesDataset(config.data, config.params)
    dataset.init_multiple_merge()
    return split_dataset(dataset, args, config.data)
