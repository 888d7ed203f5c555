"""Readers for the on-disk region-feature format of the reference (`utils/dataset/features_reader.py:16-188`): an LMDB whose values are
pickled dicts holding fp32 arrays either as raw bytes (`feature` / `bbox` / `cls_prob` / `image_width` / `image_height`, the "old"
convention) or as base64 text (`features` / `boxes` / `cls_prob` / `image_w` / `image_h`), plus a pickled key list under `b"keys"`.

    reader = BnBFeaturesReader(path)            # keys "listing-photo";  YTbFeaturesReader: keys "video/frame"
    features, locations, probs = reader[("123-456", "123-789")]

returns what the reference returns for a trajectory of photos: the regions of all photos concatenated, a leading global region (mean
feature, whole-image box, uniform class distribution), boxes normalised to [0, 1] with the relative area in column 4 and ones in
the orientation columns 5-10 (`_get_boxes`, `_get_locations`, `__getitem__`, lines 91-179).  Host-side numpy: this is the loader's
side of the batch tuple; `ytvln.batch` takes it from there on the device.

Backends: a directory path (needs the `lmdb` package -- it is not a dependency of the GPU path and is imported lazily), an object with
LMDB's `begin()` / `get()` transaction interface, or a plain `dict` mapping key bytes to value bytes (tests, other stores).
"""
from __future__ import annotations

import base64
import pickle
from dataclasses import dataclass
from pathlib import Path
from typing import Dict, List, Sequence, Tuple, Union

import numpy as np

FEATURE_DIM, NUM_CLASSES = 2048, 1601


@dataclass
class Record:
    photo_id: object
    listing_id: object
    num_boxes: int
    image_width: int
    image_height: int
    cls_prob: np.ndarray
    features: np.ndarray
    boxes: np.ndarray


class _DictTxn:
    def __init__(self, d):
        self._d = d

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def get(self, key):
        return self._d.get(bytes(key))


class _DictEnv:
    def __init__(self, d):
        self._d = d

    def begin(self, write=False, buffers=False):
        return _DictTxn(self._d)


def _open(source):
    if isinstance(source, dict):
        return _DictEnv(source)
    if hasattr(source, "begin"):
        return source
    try:
        import lmdb
    except ImportError as e:       # pragma: no cover - depends on the environment
        raise RuntimeError("reading an LMDB directory needs the `lmdb` package; pass an opened environment or a dict instead") from e
    return lmdb.open(str(source), readonly=True, readahead=False, max_readers=20, lock=False, map_size=int(1e9))


class FeaturesReader:
    """features_reader.py:16-66: key index over one or several stores; `__getitem__(keys)` -> the unpickled items."""

    def __init__(self, path: Union[Path, str, dict, object, Sequence]):
        if isinstance(path, (Path, str, dict)) or hasattr(path, "begin"):
            path = [path]
        self.envs = [_open(p) for p in path]
        self.keys: Dict[str, int] = {}
        for i, env in enumerate(self.envs):
            with env.begin(write=False, buffers=True) as txn:
                bkeys = txn.get("keys".encode())
                if bkeys is None:
                    raise RuntimeError("Please preload keys in the LMDB")
                for k in pickle.loads(bytes(bkeys)):
                    self.keys[k.decode()] = i

    def __len__(self):
        return len(self.keys)

    def _items(self, keys: Tuple) -> List:
        for key in keys:
            if not isinstance(key, str) or key not in self.keys:
                raise TypeError(f"invalid key: {key}")
        env_idx = [self.keys[key] for key in keys]
        items = [None] * len(keys)
        for idx in set(env_idx):                      # one transaction per store
            with self.envs[idx].begin(write=False) as txn:
                for i, (idx_i, key) in enumerate(zip(env_idx, keys)):
                    if idx_i != idx:
                        continue
                    item = txn.get(key.encode())
                    if item is not None:
                        items[i] = pickle.loads(bytes(item))
        return items

    def __getitem__(self, keys: Tuple) -> List:
        return self._items(keys)


def decode_record(item: Dict, photo_id=None, listing_id=None) -> Record:
    """features_reader.py:124-150 (`_convert_item`): both field conventions."""
    old = "image_width" in item
    image_w = int(item["image_width" if old else "image_w"])
    image_h = int(item["image_height" if old else "image_h"])
    features = np.frombuffer(item["feature"] if old else base64.b64decode(item["features"]), dtype=np.float32).reshape((-1, FEATURE_DIM))
    boxes = np.frombuffer(item["bbox"] if old else base64.b64decode(item["boxes"]), dtype=np.float32).reshape((-1, 4))
    cls_prob = np.frombuffer(item["cls_prob"] if old else base64.b64decode(item["cls_prob"]), dtype=np.float32).reshape((-1, NUM_CLASSES))
    return Record(photo_id, listing_id, int(boxes.shape[0]), image_w, image_h, cls_prob, features, boxes)


def normalise_boxes(record: Record) -> np.ndarray:
    """features_reader.py:91-107 (`_get_boxes`): x1/w, y1/h, x2/w, y2/h, area / (w h), fp32."""
    boxes = record.boxes
    area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    area = area / (record.image_width * record.image_height)
    out = np.zeros((len(boxes), 5), dtype=np.float32)
    out[:, 0] = boxes[:, 0] / record.image_width
    out[:, 1] = boxes[:, 1] / record.image_height
    out[:, 2] = boxes[:, 2] / record.image_width
    out[:, 3] = boxes[:, 3] / record.image_height
    out[:, 4] = area
    return out


def locations_of(boxes5: np.ndarray) -> np.ndarray:
    """features_reader.py:110-124 (`_get_locations`): 11 columns, the orientation columns stay 1."""
    loc = np.ones((len(boxes5), 11), dtype=np.float32)
    loc[:, :5] = boxes5[:, :5]
    return loc


class BaseFeaturesReader(FeaturesReader):
    def _split_key(self, key: str):
        raise NotImplementedError("_split_key: not implemented!")

    def __getitem__(self, query: Tuple):
        """features_reader.py:153-179."""
        l_boxes, l_probs, l_features = [], [], []
        for key, item in zip(list(query), self._items(query)):
            photo_id, listing_id = self._split_key(key)
            record = decode_record(item, photo_id, listing_id)
            l_boxes.append(normalise_boxes(record))
            l_probs.append(record.cls_prob)
            l_features.append(record.features)
        features = np.concatenate(l_features, axis=0)
        boxes = np.concatenate(l_boxes, axis=0)
        probs = np.concatenate(l_probs, axis=0)
        locations = locations_of(boxes)
        if features.size == 0:
            raise RuntimeError("Features could not be correctly read")
        g_feature = features.mean(axis=0, keepdims=True)                      # the global region
        g_location = np.array([[0, 0, 1, 1, 1, 0, 1, 0, 1, 0, 1]])
        g_prob = np.ones(shape=(1, NUM_CLASSES)) / NUM_CLASSES
        return (np.concatenate([g_feature, features], axis=0), np.concatenate([g_location, locations], axis=0),
                np.concatenate([g_prob, probs], axis=0))


class BnBFeaturesReader(BaseFeaturesReader):
    def _split_key(self, key: str):
        return map(int, key.split("-"))


class YTbFeaturesReader(BaseFeaturesReader):
    def _split_key(self, key: str):
        return key.split("/")


# ------------------------------------------------------------------------------------------------------------------
# Room-to-Room panoramas (features_reader.py:193-341): one record per viewpoint, keys "scan-viewpoint"; every region also carries a
# heading / elevation, and the location vector encodes them relative to the agent's current and next heading.
# ------------------------------------------------------------------------------------------------------------------
_PANO_ARRAYS = (("features", (-1, FEATURE_DIM)), ("boxes", (-1, 4)), ("cls_prob", (-1, NUM_CLASSES)), ("viewHeading", None),
                ("viewElevation", None), ("featureHeading", None), ("featureElevation", None), ("featureViewIndex", None))


def decode_pano_record(item: Dict) -> Dict:
    """features_reader.py:193-236 (`_convert_item`): base64 fields -> fp32 arrays, sizes -> ints."""
    out = dict(item)
    for k in ("image_w", "image_h", "vfov"):
        out[k] = int(item[k])
    for k, shape in _PANO_ARRAYS:
        a = np.frombuffer(base64.b64decode(item[k]), dtype=np.float32)
        out[k] = a.reshape(shape) if shape else a
    return out


def pano_locations(boxes5: np.ndarray, feat_headings: np.ndarray, feat_elevations: np.ndarray, heading: float, next_heading: float) -> np.ndarray:
    """features_reader.py:258-282 (`_get_locations`)."""
    loc = np.ones((len(boxes5), 11), dtype=np.float32)
    loc[:, :5] = boxes5[:, :5]
    loc[:, 5] = np.sin(feat_headings - heading)
    loc[:, 6] = np.cos(feat_headings - heading)
    loc[:, 7] = np.sin(feat_elevations)
    loc[:, 8] = np.cos(feat_elevations)
    loc[:, 9] = np.sin(feat_headings - next_heading)
    loc[:, 10] = np.cos(feat_headings - next_heading)
    return loc


class PanoFeaturesReader(FeaturesReader):
    """features_reader.py:285-341: `reader[(key, heading, next_heading)]` -> (features, locations, probs) with the global region first."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.viewpoints: Dict[str, set] = {}
        for key in self.keys:
            scan_id, viewpoint_id = key.split("-")
            self.viewpoints.setdefault(scan_id, set()).add(viewpoint_id)

    def __getitem__(self, query: Tuple):
        key, heading, next_heading = query
        if key not in self.keys:
            raise TypeError(f"invalid key: {key}")
        with self.envs[self.keys[key]].begin(write=False) as txn:
            item = decode_pano_record(pickle.loads(bytes(txn.get(key.encode()))))
        rec = Record(None, None, len(item["boxes"]), item["image_w"], item["image_h"], item["cls_prob"], item["features"], item["boxes"])
        features, probs = item["features"], item["cls_prob"]
        locations = pano_locations(normalise_boxes(rec), item["featureHeading"], item["featureElevation"], heading, next_heading)
        g_feature = features.mean(axis=0, keepdims=True)
        g_location = np.array([[0, 0, 1, 1, 1, np.sin(0 - heading), np.cos(0 - heading), np.sin(0), np.cos(0),
                                np.sin(0 - next_heading), np.cos(0 - next_heading)]])
        g_prob = np.ones(shape=(1, NUM_CLASSES)) / NUM_CLASSES
        return (np.concatenate([g_feature, features], axis=0), np.concatenate([g_location, locations], axis=0),
                np.concatenate([g_prob, probs], axis=0))
