"""Bank build + persistence.

Reference dataflow (mtgvision/qdrant_populate.py:70-90, four worker processes, encoder batch = 1):
    retrieve existing ids -> for each missing card: download -> imread -> make_cropped(size_hw=encoder.input_hwc)
    -> encoder.predict -> QdrantPoint(id, vector) -> save_points.
Here the same steps run per batch on the GPU: one ragged `make_cropped` launch, one batched encoder forward,
one append to the device-resident bank.  Downloading / decoding stays with the caller (images arrive as arrays).
"""

from __future__ import annotations

import json
import os
from typing import Iterable, List, Sequence, Tuple

import numpy as np
import torch

from . import native
from .adapters import QdrantPoint, VectorStoreQdrant
from .encoder import Encoder


def make_cropped(images: Sequence[np.ndarray], size_hw=(192, 128), device=None) -> torch.Tensor:
    """list of uint8 HWC card images (any sizes) -> (n, h, w, 3) float32 in [0,1] on the GPU
    (SyntheticBgFgMtgImages.make_cropped, encoder_datasets.py:733-753)."""
    native.require_gpu()
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    n = len(images)
    oh, ow = size_hw
    out = torch.empty((n, oh, ow, 3), dtype=torch.float32, device=dev)
    if n == 0:
        return out
    flat, offs, hw, pos = [], [], [], 0
    for im in images:
        im = np.asarray(im)
        assert im.ndim == 3 and im.shape[-1] == 3 and im.dtype == np.uint8, f"{im.shape} {im.dtype}"
        flat.append(np.ascontiguousarray(im).reshape(-1))
        offs.append(pos)
        hw.append(im.shape[:2])
        pos += im.size
    buf = torch.from_numpy(np.concatenate(flat)).to(dev)
    offs_t = torch.tensor(offs, dtype=torch.int64, device=dev)
    hw_t = torch.tensor(hw, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        native.check(native.lib().mtgv_make_cropped(native.ptr(buf), native.ptr(offs_t), native.ptr(hw_t), n, oh, ow, native.ptr(out), native.stream()))
    return out


def build_bank(cards: Iterable[Tuple[str, np.ndarray]], encoder: Encoder, store: VectorStoreQdrant, batch_size: int = 256) -> int:
    """cards: (id, uint8 HWC image) pairs.  Embeds every card whose id is not in the store yet and upserts it.
    Returns the number of cards added (CardProcessor.process_batch returns the same count)."""
    h, w, _ = encoder.input_hwc
    added = 0
    batch: List[Tuple[str, np.ndarray]] = []

    def flush():
        nonlocal added, batch
        if not batch:
            return
        existing = {p.id for p in store.retrieve((cid for cid, _ in batch), with_payload=False)}
        todo = [(cid, im) for cid, im in batch if str(cid) not in existing]
        batch = []
        if not todo:
            return
        x = make_cropped([im for _, im in todo], (h, w), encoder.device)
        z = encoder.encode(x).cpu().numpy()
        store.save_points(QdrantPoint(id=str(cid), vector=zi, payload=None) for (cid, _), zi in zip(todo, z))
        added += len(todo)

    for item in cards:
        batch.append(item)
        if len(batch) >= batch_size:
            flush()
    flush()
    return added


def save_store(store: VectorStoreQdrant, path: str) -> None:
    """On-disk bank: <path>.npz (ids as fixed-width strings + float32 matrix of the stored, normalised vectors)
    and <path>.payload.json."""
    n = len(store._ids)
    vecs = store._bank.rows(0, n) if n else np.zeros((0, store._VECTOR_SIZE), np.float32)
    np.savez(path + ".npz", ids=np.asarray(store._ids, dtype=np.str_), vectors=vecs)
    with open(path + ".payload.json", "w") as f:
        json.dump({k: v for k, v in store._payload.items()}, f)


def load_store(path: str, capacity: int = 131072) -> VectorStoreQdrant:
    d = np.load(path + ".npz", allow_pickle=False)
    ids, vecs = [str(i) for i in d["ids"]], d["vectors"]
    store = VectorStoreQdrant(capacity=max(capacity, len(ids)))
    payload = {}
    if os.path.exists(path + ".payload.json"):
        payload = json.load(open(path + ".payload.json"))
    for s in range(0, len(ids), 4096):
        store.save_points(QdrantPoint(id=i, vector=v, payload=payload.get(i)) for i, v in zip(ids[s : s + 4096], vecs[s : s + 4096]))
    return store
