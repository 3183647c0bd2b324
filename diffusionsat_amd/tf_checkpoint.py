"""Reader for TensorFlow 2 object-graph checkpoints (TensorBundle format), pure Python.

The reference restores its model with ``tf.train.Checkpoint(step, model).restore(latest)``
(satuniformity/DiffusionSampler.py:215-227).  TensorFlow is not a dependency of this package,
so this module parses the two files of a checkpoint directly:

  ``<prefix>.index``                 an SSTable ("leveldb table" format, no compression as
                                     written by tensorflow/core/util/tensor_bundle) mapping
                                     tensor keys to ``BundleEntryProto`` records
  ``<prefix>.data-00000-of-00001``   the raw little-endian tensor bytes

and returns the five MLPs' ``kernel``/``bias`` arrays.  Expected variable keys (derived from
the attribute names of model/query_sat.py:117-122 and model/mlp.py:24,37-39):

  ``model/<mlp>/dense_layers/<i>/{kernel,bias}/.ATTRIBUTES/VARIABLE_VALUE``

UNVERIFIED AGAINST A TENSORFLOW-WRITTEN FILE -- and not verifiable here: no checkpoint ships with
the reference (README.md:38-40 points to GitHub Releases) and TensorFlow is not installed, so the
format knowledge is from the published TensorBundle / leveldb table layouts.  The tests read files
produced by ``write_checkpoint`` below and by an independently written table writer in
tests/test_tf_checkpoint.py that exercises what TensorFlow's own writer does and ours does not:
prefix-shared keys between restart points, several data blocks, a second data shard, entries of
other dtypes (the object graph string, optimizer slots) and missing model keys.  The ``.npz`` format
of ``diffusionsat_amd.weights`` stays the documented, tested path; ``tools/export_tf_checkpoint.py``
converts a checkpoint to it on a machine that has TensorFlow.
"""
from __future__ import annotations

import os
import re
import struct
from typing import Dict, Optional, Tuple

import numpy as np

from .weights import MLP_NAMES, WeightSet, layer_dims

TABLE_MAGIC = 0xDB4775248B80FB57
FOOTER_LEN = 48
DT_FLOAT, DT_INT64 = 1, 9
KEY_SUFFIX = "/.ATTRIBUTES/VARIABLE_VALUE"


# ------------------------------------------------------------------------------- varints / protobuf
def _get_varint(buf: bytes, pos: int) -> Tuple[int, int]:
    shift = result = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _put_varint(v: int) -> bytes:
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _parse_proto(buf: bytes) -> Dict[int, list]:
    """Minimal protobuf wire parser: {field: [values]} (varint -> int, length-delimited -> bytes,
    fixed32/64 -> int)."""
    out: Dict[int, list] = {}
    pos = 0
    while pos < len(buf):
        tag, pos = _get_varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            ln, pos = _get_varint(buf, pos)
            v = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        out.setdefault(field, []).append(v)
    return out


def _parse_entry(value: bytes):
    """BundleEntryProto: dtype=1, shape=2 (TensorShapeProto: dim=2 {size=1}), shard_id=3, offset=4, size=5."""
    p = _parse_proto(value)
    dtype = p.get(1, [0])[0]
    shape = []
    if 2 in p:
        for dim in _parse_proto(p[2][0]).get(2, []):
            shape.append(_parse_proto(dim).get(1, [0])[0])
    return dtype, tuple(shape), p.get(3, [0])[0], p.get(4, [0])[0], p.get(5, [0])[0]


# ------------------------------------------------------------------------------- leveldb table
def _read_block(data: bytes, offset: int, size: int):
    block = data[offset:offset + size]
    ctype = data[offset + size]
    if ctype != 0:
        raise ValueError("compressed index blocks (type %d) are not supported" % ctype)
    n_restarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    limit = len(block) - 4 - 4 * n_restarts
    pos, key = 0, b""
    while pos < limit:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        yield key, block[pos:pos + vlen]
        pos += vlen


def read_header(index_path: str) -> Dict[str, int]:
    """BundleHeaderProto of the index (the entry under the empty key): num_shards=1, endianness=2."""
    hdr = {"num_shards": 1, "endianness": 0}
    for key, value in _iter_table(index_path):
        if key == b"":
            p = _parse_proto(value)
            hdr["num_shards"] = p.get(1, [1])[0]
            hdr["endianness"] = p.get(2, [0])[0]
        break       # keys are sorted: the header, if present, is first
    return hdr


def _iter_table(index_path: str):
    with open(index_path, "rb") as f:
        data = f.read()
    if len(data) < FOOTER_LEN:
        raise ValueError("%s: too short for a table footer" % index_path)
    footer = data[-FOOTER_LEN:]
    if struct.unpack_from("<Q", footer, FOOTER_LEN - 8)[0] != TABLE_MAGIC:
        raise ValueError("%s: not a TensorBundle index (bad magic)" % index_path)
    pos = 0
    _, pos = _get_varint(footer, pos)   # metaindex handle
    _, pos = _get_varint(footer, pos)
    idx_off, pos = _get_varint(footer, pos)
    idx_size, pos = _get_varint(footer, pos)
    for _, handle in _read_block(data, idx_off, idx_size):
        boff, p2 = _get_varint(handle, 0)
        bsize, _ = _get_varint(handle, p2)
        for key, value in _read_block(data, boff, bsize):
            yield key, value


def read_index(index_path: str) -> Dict[str, tuple]:
    entries = {}
    for key, value in _iter_table(index_path):
        if key == b"":
            continue   # BundleHeaderProto
        entries[key.decode("utf-8")] = _parse_entry(value)
    return entries


def latest_checkpoint(model_path: str) -> Optional[str]:
    """``tf.train.CheckpointManager.latest_checkpoint``: the ``checkpoint`` text file names it."""
    if os.path.isfile(model_path + ".index"):
        return model_path
    state = os.path.join(model_path, "checkpoint")
    if os.path.isfile(state):
        with open(state, "r") as f:
            for line in f:
                m = re.match(r'\s*model_checkpoint_path:\s*"(.*)"', line)
                if m:
                    p = m.group(1)
                    return p if os.path.isabs(p) else os.path.join(model_path, p)
    if os.path.isdir(model_path):
        cands = [f[:-6] for f in os.listdir(model_path) if f.endswith(".index")]
        if cands:
            def num(s):
                m = re.search(r"(\d+)$", s)
                return int(m.group(1)) if m else -1
            return os.path.join(model_path, max(cands, key=num))
    return None


def read_tensors(prefix: str, wanted=None) -> Dict[str, np.ndarray]:
    entries = read_index(prefix + ".index")
    hdr = read_header(prefix + ".index")
    if hdr["endianness"] != 0:
        raise ValueError("%s: big-endian tensor bundle" % prefix)
    out = {}
    shards = {}
    for key, (dtype, shape, shard, offset, size) in entries.items():
        if wanted is not None and key not in wanted:
            continue
        if dtype not in (DT_FLOAT, DT_INT64):
            continue
        if shard not in shards:
            n = max(hdr["num_shards"], 1 + max(e[2] for e in entries.values()))
            shards[shard] = np.memmap("%s.data-%05d-of-%05d" % (prefix, shard, n), dtype=np.uint8, mode="r")
        raw = bytes(shards[shard][offset:offset + size])
        arr = np.frombuffer(raw, dtype="<f4" if dtype == DT_FLOAT else "<i8")
        out[key] = arr.reshape(shape).copy()
    return out


last_restore_report: Dict[str, list] = {"restored": [], "missing": [], "unused": []}


def load_queriesat_weights(model_path: str, layout=None, base: Optional[WeightSet] = None) -> Optional[WeightSet]:
    """Weights of the five MLPs from the latest checkpoint under ``model_path`` (or None if there
    is none).  The reference restores with ``.expect_partial()`` (satuniformity/DiffusionSampler.py:220):
    keys of the file the model does not have (optimizer slots, ``step``, the object graph) are ignored,
    and model variables the file lacks silently keep their initial values.  The same here when ``base``
    (the model's freshly initialised weights) is given -- ``last_restore_report`` then lists what was
    restored, what stayed at its initial value and which float tensors of the file went unused; without
    ``base`` a missing model variable raises KeyError."""
    prefix = latest_checkpoint(model_path) if model_path else None
    if prefix is None:
        return None
    tensors = read_tensors(prefix)
    report = {"restored": [], "missing": [], "unused": []}
    ws: WeightSet = {}
    for name, dims in layer_dims(layout):
        layers = []
        for i, (fi, fo) in enumerate(dims):
            kk = "model/%s/dense_layers/%d/kernel%s" % (name, i, KEY_SUFFIX)
            bk = "model/%s/dense_layers/%d/bias%s" % (name, i, KEY_SUFFIX)
            if kk not in tensors or bk not in tensors:
                if base is None:
                    raise KeyError("checkpoint %s lacks %s (found %d tensors)" % (prefix, kk, len(tensors)))
                bw, bb = base[name][i]
                w = tensors.get(kk, np.asarray(bw, np.float32))
                b = tensors.get(bk, np.asarray(bb, np.float32))
                report["missing"] += [k for k in (kk, bk) if k not in tensors]
                report["restored"] += [k for k in (kk, bk) if k in tensors]
            else:
                w, b = tensors[kk], tensors[bk]
                report["restored"] += [kk, bk]
            if w.shape != (fi, fo) or b.shape != (fo,):
                raise ValueError("%s: kernel %s / bias %s, expected (%d,%d)/(%d,)" % (kk, w.shape, b.shape, fi, fo, fo))
            layers.append((w.astype(np.float32), b.astype(np.float32)))
        ws[name] = layers
    used = set(report["restored"])
    report["unused"] = sorted(k for k in tensors if k not in used)
    last_restore_report.clear()
    last_restore_report.update(report)
    # expect_partial() tolerates gaps, but a file that fills none of an MLP's variables is not a QuerySAT
    # checkpoint under these key names (they are derived from the source, SURVEY.md §5: unverified against a
    # TensorFlow-written file): report it instead of sampling with fresh weights that claim to be restored
    wholly_missing = [name for name, dims in layer_dims(layout)
                      if not any(k.startswith("model/%s/" % name) for k in report["restored"])]
    if wholly_missing:
        raise KeyError("checkpoint %s holds no variables of %s under model/<mlp>/dense_layers/<i>/{kernel,bias} "
                       "(%d tensors in the file; first keys: %s)"
                       % (prefix, ", ".join(wholly_missing), len(tensors), sorted(tensors)[:4]))
    if report["missing"]:
        import warnings
        warnings.warn("checkpoint %s: %d model variable(s) keep their initial values: %s"
                      % (prefix, len(report["missing"]), ", ".join(report["missing"])))
    return ws


# ------------------------------------------------------------------------------- writer (tests, export)
def _entry_proto(dtype: int, shape, offset: int, size: int) -> bytes:
    dims = b"".join(b"\x12" + _put_varint(len(d)) + d for d in (b"\x08" + _put_varint(int(s)) for s in shape))
    out = b"\x08" + _put_varint(dtype) + b"\x12" + _put_varint(len(dims)) + dims
    out += b"\x20" + _put_varint(offset) + b"\x28" + _put_varint(size)
    return out


def _block(entries) -> bytes:
    """leveldb table block: entries (no prefix sharing: a restart point at every entry), the
    restart offsets (uint32 each) and their count."""
    body = bytearray()
    restarts = []
    for key, value in entries:
        restarts.append(len(body))
        body += _put_varint(0) + _put_varint(len(key)) + _put_varint(len(value)) + key + value
    if not restarts:
        restarts = [0]
    for r in restarts:
        body += struct.pack("<I", r)
    body += struct.pack("<I", len(restarts))
    return bytes(body)


def write_checkpoint(directory: str, ws: WeightSet, step: int = 1) -> str:
    """Write a TensorBundle checkpoint holding the weight set (same key layout the reference's
    checkpoints use).  For tests and for exporting weights; CRCs are left zero."""
    os.makedirs(directory, exist_ok=True)
    prefix = os.path.join(directory, "ckpt-%d" % step)
    tensors = {}
    for name in MLP_NAMES:
        for i, (w, b) in enumerate(ws[name]):
            tensors["model/%s/dense_layers/%d/kernel%s" % (name, i, KEY_SUFFIX)] = np.asarray(w, dtype="<f4")
            tensors["model/%s/dense_layers/%d/bias%s" % (name, i, KEY_SUFFIX)] = np.asarray(b, dtype="<f4")
    tensors["step" + KEY_SUFFIX] = np.asarray(step, dtype="<i8")
    data = bytearray()
    entries = [(b"", b"\x08\x01")]   # BundleHeaderProto{num_shards: 1}
    for key in sorted(tensors):
        arr = tensors[key]
        raw = arr.tobytes()
        entries.append((key.encode(), _entry_proto(DT_INT64 if arr.dtype.kind == "i" else DT_FLOAT, arr.shape,
                                                   len(data), len(raw))))
        data += raw
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        f.write(bytes(data))
    blk = _block(entries)
    out = bytearray(blk) + b"\x00" + struct.pack("<I", 0)          # data block + trailer
    meta_off = len(out)
    meta = _block([])
    out += meta + b"\x00" + struct.pack("<I", 0)
    idx_off = len(out)
    handle = _put_varint(0) + _put_varint(len(blk))
    idx = _block([(entries[-1][0] + b"\xff", handle)])
    out += idx + b"\x00" + struct.pack("<I", 0)
    footer = _put_varint(meta_off) + _put_varint(len(meta)) + _put_varint(idx_off) + _put_varint(len(idx))
    footer += b"\x00" * (FOOTER_LEN - 8 - len(footer)) + struct.pack("<Q", TABLE_MAGIC)
    out += footer
    with open(prefix + ".index", "wb") as f:
        f.write(bytes(out))
    with open(os.path.join(directory, "checkpoint"), "w") as f:
        f.write('model_checkpoint_path: "ckpt-%d"\nall_model_checkpoint_paths: "ckpt-%d"\n' % (step, step))
    return prefix
