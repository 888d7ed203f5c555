"""Golden vectors for the on-disk feature format: synthetic records in both field conventions are served to the REFERENCE's
`BnBFeaturesReader` / `YTbFeaturesReader` (utils/dataset/features_reader.py) through a dict-backed stand-in for the lmdb environment
(the `lmdb` package is absent from this image; only its open/begin/get calls are emulated, the reference's decoding runs unchanged).

    python oracle/gen_golden_features.py      # writes tests/golden/g8_features.npz   (TEST INFRASTRUCTURE; needs /root/reference)
"""
import base64
import os
import pickle
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

ref_import.install_stubs()
sys.path.insert(0, ref_import.REFERENCE_ROOT)

STORES = {}


class Txn:
    def __init__(self, d):
        self.d = d

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def get(self, k):
        return self.d.get(bytes(k))


class Env:
    def __init__(self, d):
        self.d = d

    def begin(self, write=False, buffers=False):
        return Txn(self.d)


sys.modules["lmdb"].open = lambda path, **kw: Env(STORES[path])
import utils.dataset.features_reader as FR  # noqa: E402

rs = np.random.RandomState(11)


def record(nb, old):
    w, h = int(rs.randint(300, 900)), int(rs.randint(300, 900))
    feat = np.maximum(rs.standard_normal((nb, 2048)), 0).astype(np.float32)
    xy = np.sort(rs.uniform(0, 1, (nb, 2, 2)), axis=-1)
    boxes = np.stack([xy[:, 0, 0] * w, xy[:, 1, 0] * h, xy[:, 0, 1] * w, xy[:, 1, 1] * h], 1).astype(np.float32)
    logits = rs.standard_normal((nb, 1601)).astype(np.float32)
    prob = np.exp(logits - logits.max(-1, keepdims=True))
    prob = (prob / prob.sum(-1, keepdims=True)).astype(np.float32)
    if old:
        return {"image_width": w, "image_height": h, "feature": feat.tobytes(), "bbox": boxes.tobytes(), "cls_prob": prob.tobytes()}
    return {"image_w": w, "image_h": h, "features": base64.b64encode(feat.tobytes()), "boxes": base64.b64encode(boxes.tobytes()),
            "cls_prob": base64.b64encode(prob.tobytes())}


def store(keys, old):
    d = {k.encode(): pickle.dumps(record(int(rs.randint(3, 9)), old)) for k in keys}
    d[b"keys"] = pickle.dumps([k.encode() for k in keys])
    return d


bnb_keys = ["12-1", "12-7", "98-3"]
ytb_keys = ["vidA/000010", "vidA/000250", "vidB/000031"]
STORES["bnb_old"], STORES["bnb_new"], STORES["ytb_new"] = store(bnb_keys[:2], True), store(bnb_keys[2:], False), store(ytb_keys, False)
out = {}
r = FR.BnBFeaturesReader(["bnb_old", "bnb_new"])
out["bnb_f"], out["bnb_l"], out["bnb_p"] = r[("12-7", "98-3", "12-1")]
r = FR.YTbFeaturesReader("ytb_new")
out["ytb_f"], out["ytb_l"], out["ytb_p"] = r[("vidB/000031", "vidA/000010")]


def pano_record(nb):
    r = record(nb, False)
    r["vfov"] = 60
    for k, n in (("viewHeading", 36), ("viewElevation", 36), ("featureHeading", nb), ("featureElevation", nb), ("featureViewIndex", nb)):
        r[k] = base64.b64encode(rs.uniform(-3.1, 3.1, n).astype(np.float32).tobytes())
    return r


pano_keys = ["scanA-vp1", "scanA-vp2", "scanB-vp9"]
STORES["pano"] = {k.encode(): pickle.dumps(pano_record(int(rs.randint(3, 9)))) for k in pano_keys}
STORES["pano"][b"keys"] = pickle.dumps([k.encode() for k in pano_keys])
r = FR.PanoFeaturesReader("pano")
out["pano_f"], out["pano_l"], out["pano_p"] = r[("scanA-vp2", 0.7, -1.9)]
out["pano_viewpoints"] = np.array(sorted(f"{s}:{v}" for s, vs in r.viewpoints.items() for v in vs))
blob = {name: np.frombuffer(pickle.dumps(d), dtype=np.uint8) for name, d in STORES.items()}
np.savez_compressed(os.path.join(os.path.dirname(HERE), "tests", "golden", "g8_features.npz"), **out, **{"store_" + k: v for k, v in blob.items()})
print("wrote g8_features.npz", {k: v.shape for k, v in out.items()})
