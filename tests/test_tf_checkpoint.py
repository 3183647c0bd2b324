"""The TensorBundle reader against files written by a SECOND, independently written table writer that
does what TensorFlow's writer does and the package's own ``write_checkpoint`` does not: prefix-shared
keys between restart points (leveldb block format, restart interval 16), several data blocks behind a
multi-entry index block, tensors spread over two data shards, crc fields, entries the model does not
use (object graph string, optimizer slots, save counter) and missing model variables
(``expect_partial`` semantics, satuniformity/DiffusionSampler.py:220).  No TensorFlow-written file
exists in the reference or the container, so this is as far as the reader can be pinned here."""
import os
import struct

import numpy as np
import pytest

from diffusionsat_amd import tf_checkpoint as T
from diffusionsat_amd import weights as W

SUF = T.KEY_SUFFIX


def _varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _block_shared(entries, restart_interval=16):
    """leveldb table block WITH key prefix compression."""
    body, restarts, last = bytearray(), [], b""
    for i, (key, value) in enumerate(entries):
        if i % restart_interval == 0:
            restarts.append(len(body))
            shared = 0
        else:
            shared = 0
            while shared < min(len(key), len(last)) and key[shared] == last[shared]:
                shared += 1
        body += _varint(shared) + _varint(len(key) - shared) + _varint(len(value)) + key[shared:] + value
        last = key
    for r in restarts or [0]:
        body += struct.pack("<I", r)
    body += struct.pack("<I", len(restarts) or 1)
    return bytes(body)


def _entry(dtype, shape, shard, offset, size, crc=0x12345678):
    dims = b"".join(b"\x12" + _varint(len(d)) + d for d in (b"\x08" + _varint(int(x)) for x in shape))
    out = b"\x08" + _varint(dtype) + b"\x12" + _varint(len(dims)) + dims
    if shard:
        out += b"\x18" + _varint(shard)
    out += b"\x20" + _varint(offset) + b"\x28" + _varint(size)
    out += b"\x35" + struct.pack("<I", crc)                      # field 6, fixed32 (crc32c)
    return out


def write_like_tensorflow(directory, tensors, step=7, n_shards=2, block_entries=5, strings=None):
    """`tensors`: {key: float32/int64 array}.  Keys sorted; entries cut into data blocks of `block_entries`;
    tensors alternate between `n_shards` data files."""
    os.makedirs(directory, exist_ok=True)
    prefix = os.path.join(directory, "ckpt-%d" % step)
    datas = [bytearray() for _ in range(n_shards)]
    items = [(b"", b"\x08" + _varint(n_shards) + b"\x10\x00")]    # BundleHeaderProto{num_shards, LITTLE}
    for i, key in enumerate(sorted(list(tensors) + list(strings or {}))):
        shard = i % n_shards
        if strings and key in strings:
            raw = strings[key]
            ent = _entry(7, (), shard, len(datas[shard]), len(raw))          # DT_STRING
        else:
            arr = tensors[key]
            raw = arr.tobytes()
            ent = _entry(T.DT_INT64 if arr.dtype.kind == "i" else T.DT_FLOAT, arr.shape, shard, len(datas[shard]),
                         len(raw))
        datas[shard] += raw
        items.append((key.encode(), ent))
    for sh, d in enumerate(datas):
        with open("%s.data-%05d-of-%05d" % (prefix, sh, n_shards), "wb") as f:
            f.write(bytes(d))
    out, index_entries = bytearray(), []
    for b0 in range(0, len(items), block_entries):
        chunk = items[b0:b0 + block_entries]
        blk = _block_shared(chunk, restart_interval=3)
        index_entries.append((chunk[-1][0] + b"\x00", _varint(len(out)) + _varint(len(blk))))
        out += blk + b"\x00" + struct.pack("<I", 0)
    meta_off = len(out)
    meta = _block_shared([])
    out += meta + b"\x00" + struct.pack("<I", 0)
    idx_off = len(out)
    idx = _block_shared(index_entries, restart_interval=1)
    out += idx + b"\x00" + struct.pack("<I", 0)
    footer = _varint(meta_off) + _varint(len(meta)) + _varint(idx_off) + _varint(len(idx))
    footer += b"\x00" * (T.FOOTER_LEN - 8 - len(footer)) + struct.pack("<Q", T.TABLE_MAGIC)
    out += footer
    with open(prefix + ".index", "wb") as f:
        f.write(bytes(out))
    with open(os.path.join(directory, "checkpoint"), "w") as f:
        f.write('model_checkpoint_path: "ckpt-%d"\n' % step)
    return prefix


def _model_tensors(ws):
    t = {}
    for name in W.MLP_NAMES:
        for i, (w, b) in enumerate(ws[name]):
            t["model/%s/dense_layers/%d/kernel%s" % (name, i, SUF)] = np.asarray(w, "<f4")
            t["model/%s/dense_layers/%d/bias%s" % (name, i, SUF)] = np.asarray(b, "<f4")
    return t


def _same(a, b):
    return all(np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1])
               for n in W.MLP_NAMES for x, y in zip(a[n], b[n]))


def test_reads_prefix_shared_multiblock_two_shard_checkpoint(tmp_path):
    ws = W.glorot_weights(5, bias_scale=0.1)
    t = _model_tensors(ws)
    # what a training checkpoint also holds: optimizer slots next to every variable, the step, the save
    # counter and the serialized object graph (a string tensor)
    for k in list(t):
        base = k[:-len(SUF)]
        t[base + "/.OPTIMIZER_SLOT/optimizer/m" + SUF] = np.zeros_like(t[k])
        t[base + "/.OPTIMIZER_SLOT/optimizer/v" + SUF] = np.ones_like(t[k])
    t["step" + SUF] = np.asarray(167000, "<i8")
    t["save_counter" + SUF] = np.asarray(3, "<i8")
    prefix = write_like_tensorflow(str(tmp_path), t, step=167000,
                                   strings={"_CHECKPOINTABLE_OBJECT_GRAPH": b"\x0a\x05hello"})
    assert T.latest_checkpoint(str(tmp_path)) == prefix
    assert T.read_header(prefix + ".index") == {"num_shards": 2, "endianness": 0}
    idx = T.read_index(prefix + ".index")
    assert len(idx) == len(t) + 1 and {e[2] for e in idx.values()} == {0, 1}       # both shards in use
    back = T.load_queriesat_weights(str(tmp_path))
    assert _same(ws, back)
    rep = T.last_restore_report
    assert len(rep["restored"]) == 24 and rep["missing"] == []
    assert any("OPTIMIZER_SLOT" in k for k in rep["unused"]) and "step" + SUF in rep["unused"]
    assert int(T.read_tensors(prefix, wanted={"step" + SUF})["step" + SUF]) == 167000


def test_missing_model_variables_follow_expect_partial(tmp_path):
    ws = W.glorot_weights(6, bias_scale=0.1)
    t = _model_tensors(ws)
    gone = ["model/lit_mlp/dense_layers/1/kernel" + SUF, "model/variables_output/dense_layers/1/bias" + SUF]
    for k in gone:
        del t[k]
    write_like_tensorflow(str(tmp_path), t, n_shards=1, block_entries=4)
    with pytest.raises(KeyError):
        T.load_queriesat_weights(str(tmp_path))                      # strict: no base to fall back on
    base = W.glorot_weights(99)
    with pytest.warns(UserWarning, match="keep their initial values"):
        back = T.load_queriesat_weights(str(tmp_path), base=base)
    assert sorted(T.last_restore_report["missing"]) == sorted(gone)
    assert np.array_equal(back["lit_mlp"][1][0], base["lit_mlp"][1][0])           # stays at its initial value
    assert np.array_equal(back["lit_mlp"][1][1], ws["lit_mlp"][1][1])             # its bias WAS in the file
    assert np.array_equal(back["variables_output"][1][1], base["variables_output"][1][1])
    assert np.array_equal(back["update_gate"][0][0], ws["update_gate"][0][0])


def test_checkpoint_without_model_keys_is_not_a_restore(tmp_path):
    """A TensorBundle that holds none of an MLP's variables under the derived key names (another model, or a key
    layout this reader does not know) must not come back as "restored" with fresh Glorot weights behind it."""
    ws = W.glorot_weights(7)
    foreign = {k.replace("model/", "net/"): v for k, v in _model_tensors(ws).items()}
    write_like_tensorflow(str(tmp_path / "a"), foreign, n_shards=1)
    with pytest.raises(KeyError, match="holds no variables of"):
        T.load_queriesat_weights(str(tmp_path / "a"), base=W.glorot_weights(1))
    t = {k: v for k, v in _model_tensors(ws).items() if "/update_gate/" not in k}      # one MLP wholly absent
    write_like_tensorflow(str(tmp_path / "b"), t, n_shards=1)
    with pytest.raises(KeyError, match="update_gate"):
        T.load_queriesat_weights(str(tmp_path / "b"), base=W.glorot_weights(1))


def test_rejects_foreign_and_compressed_files(tmp_path):
    p = tmp_path / "x.index"
    p.write_bytes(b"\x00" * 100)
    with pytest.raises(ValueError):
        T.read_index(str(p))
    ws = W.glorot_weights(1)
    prefix = write_like_tensorflow(str(tmp_path / "c"), _model_tensors(ws), n_shards=1)
    raw = bytearray(open(prefix + ".index", "rb").read())
    # mark the first data block as compressed: its type byte follows the block (location from the index block)
    footer = bytes(raw[-T.FOOTER_LEN:])
    pos = 0
    _, pos = T._get_varint(footer, pos); _, pos = T._get_varint(footer, pos)
    idx_off, pos = T._get_varint(footer, pos); idx_size, pos = T._get_varint(footer, pos)
    _, handle = next(T._read_block(bytes(raw), idx_off, idx_size))
    boff, p2 = T._get_varint(handle, 0)
    bsize, _ = T._get_varint(handle, p2)
    raw[boff + bsize] = 1
    open(prefix + ".index", "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="compressed"):
        T.read_index(prefix + ".index")
