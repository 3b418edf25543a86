"""Compressed on-disk form of quantized layers: codebook + canonical-Huffman stream of the centroid indices.

The reference keeps nothing on disk and only names the third stage of Deep Compression (Huffman coding,
README.md:9); what ``get_quantized_weight`` returns -- ``cluster_centers_`` and ``labels_``
(neural_network_compression/common/utility.py:239) -- is exactly a codebook and an index stream, so the stored layer is
those two, with the indices entropy coded on the GPU (include/nnc.h, nnc_huffman_*; csrc/nnc_codec.hip).  Decoding gives
back ``cluster_centers_[labels_]`` bit for bit.

File layout (little endian), one record per tensor:

    "NNC1" | u32 n_tensors
    per tensor: u16 name length, name (utf-8) | u8 ndim, u64 shape[ndim] | u32 K | u8 label_bytes | u64 N | u64 total_bits
                f32 codebook[K] | u8 code_length[K] | u32 chunk_bits[nchunks]  (nchunks = ceil(N / 1024))
                u32 words[ceil(total_bits / 32)]                                  (MSB-first bit stream)
    a tensor that passed through unquantized ("not enough bits") is stored raw: K = 0, then f32 data[N].
"""
from __future__ import annotations

import ctypes
import struct
from typing import Dict, Tuple

import numpy as np
import torch

from . import _native as nat
from . import ops

MAGIC = b"NNC1"
CHUNK = 1024


def _flatten_lengths(lengths: np.ndarray, counts: np.ndarray) -> np.ndarray:
    """Code lengths beyond 32 bits (possible only for hugely skewed histograms with hundreds of symbols) are replaced by a
    fixed-width code over the used symbols: still a prefix code, decodable by the same tables."""
    if int(lengths.max(initial=0)) <= 32:
        return lengths
    used = counts > 0
    width = max(1, int(np.ceil(np.log2(max(2, int(used.sum()))))))
    out = np.zeros_like(lengths)
    out[used] = width
    return out


def encode_indices(labels: torch.Tensor, k: int, counts: np.ndarray | None = None):
    """labels: device uint8 / int16-storage centroid indices.  Returns (words uint32 device tensor, chunk_bits np.uint32[nchunks],
    lengths np.uint8[k], total_bits)."""
    L = nat.load()
    n = labels.numel()
    lb = ops._label_bytes(labels)
    if counts is None:
        counts = ops.bincount(labels, k).cpu().numpy()
    counts = np.asarray(counts, dtype=np.int64)
    lengths, _, _ = ops.huffman_lengths(counts)
    lengths = _flatten_lengths(np.ascontiguousarray(lengths, dtype=np.uint8), counts)
    codes = np.zeros(k, dtype=np.uint32)
    nat.check(L.nnc_huffman_codes(lengths.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), k, codes.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))))
    dev = labels.device
    stream = ops._stream(labels)
    len_d = torch.from_numpy(lengths).to(dev)
    codes_d = torch.from_numpy(codes.view(np.int32)).to(dev)
    nchunks = int(L.nnc_codec_chunks(n))
    off = torch.empty(nchunks + 1, dtype=torch.int64, device=dev)
    nat.check(L.nnc_huffman_chunk_offsets(ops._ptr(labels), lb, n, len_d.data_ptr(), k, off.data_ptr(), stream))
    total_bits = int((counts * lengths.astype(np.int64)).sum())     # known on the host: no read of the device total needed
    nwords = total_bits // 32 + 2
    words = torch.empty(nwords, dtype=torch.int32, device=dev)
    nat.check(L.nnc_huffman_encode(ops._ptr(labels), lb, n, codes_d.data_ptr(), len_d.data_ptr(), k, off.data_ptr(), words.data_ptr(), nwords, stream))
    off_h = off.cpu().numpy()
    assert int(off_h[-1]) == total_bits, (int(off_h[-1]), total_bits)
    chunk_bits = np.diff(off_h).astype(np.uint32)
    return words[: (total_bits + 31) // 32], chunk_bits, lengths, total_bits


def decode_indices(words: torch.Tensor, chunk_bits: np.ndarray, n: int, lengths: np.ndarray, k: int, label_bytes: int) -> torch.Tensor:
    """The inverse of encode_indices on the device; raises if the stream does not parse."""
    L = nat.load()
    dev = words.device
    stream = ops._stream(words)
    tb = int(L.nnc_huffman_decode_tables_bytes())
    tables = np.zeros(tb, dtype=np.uint8)
    lengths = np.ascontiguousarray(lengths, dtype=np.uint8)
    nat.check(L.nnc_huffman_decode_tables(lengths.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), k, tables.ctypes.data, tb))
    tables_d = torch.from_numpy(tables).to(dev)
    off = np.zeros(chunk_bits.size + 1, dtype=np.int64)
    np.cumsum(chunk_bits.astype(np.int64), out=off[1:])
    off_d = torch.from_numpy(off).to(dev)
    padded = torch.zeros(words.numel() + 2, dtype=torch.int32, device=dev)   # the decoder may look one word past the end
    padded[: words.numel()] = words
    out = torch.empty(n, dtype=torch.uint8 if label_bytes == 1 else torch.int16, device=dev)
    bad = torch.zeros(1, dtype=torch.int32, device=dev)
    nat.check(L.nnc_huffman_decode(padded.data_ptr(), off_d.data_ptr(), n, tables_d.data_ptr(), k, ops._ptr(out), label_bytes, bad.data_ptr(), stream))
    if int(bad.item()):
        raise ValueError("corrupt index stream")
    return out


def pack_tensor(name: str, shape, model, raw: torch.Tensor | None = None) -> bytes:
    """One record.  model: kmeans.QuantizedModel (or None with ``raw`` = the unquantized float32 tensor)."""
    nm = name.encode("utf-8")
    head = struct.pack("<H", len(nm)) + nm + struct.pack("<B", len(shape)) + b"".join(struct.pack("<Q", int(d)) for d in shape)
    n = int(np.prod(shape)) if len(shape) else 1
    if model is None:
        data = np.ascontiguousarray(raw.detach().cpu().numpy(), dtype=np.float32).reshape(-1)
        return head + struct.pack("<IBQQ", 0, 0, n, 0) + data.tobytes()
    k = int(model.cluster_centers_.size)
    labels = model.labels_compact_
    counts = model.counts_device_.cpu().numpy() if getattr(model, "counts_device_", None) is not None else None
    words, chunk_bits, lengths, total_bits = encode_indices(labels, k, counts)
    lb = ops._label_bytes(labels)
    body = struct.pack("<IBQQ", k, lb, n, total_bits)
    body += np.ascontiguousarray(model.cluster_centers_.ravel(), dtype=np.float32).tobytes()
    body += lengths.tobytes() + chunk_bits.tobytes() + words.cpu().numpy().tobytes()
    return head + body


def save_compressed(path: str, tensors: Dict[str, Tuple[tuple, object, torch.Tensor | None]]) -> int:
    """tensors: name -> (shape, QuantizedModel | None, raw tensor if unquantized).  Returns the file size in bytes."""
    blob = MAGIC + struct.pack("<I", len(tensors))
    for name, (shape, model, raw) in tensors.items():
        blob += pack_tensor(name, tuple(shape), model, raw)
    with open(path, "wb") as f:
        f.write(blob)
    return len(blob)


def load_compressed(path: str, device=None) -> Dict[str, torch.Tensor]:
    """name -> decoded float32 tensor on the device: cluster_centers_[labels_] in the stored shape."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    with open(path, "rb") as f:
        blob = f.read()
    if blob[:4] != MAGIC:
        raise ValueError("not an NNC1 file")
    (nt,) = struct.unpack_from("<I", blob, 4)
    pos = 8
    out = {}
    for _ in range(nt):
        (ln,) = struct.unpack_from("<H", blob, pos); pos += 2
        name = blob[pos: pos + ln].decode("utf-8"); pos += ln
        (nd,) = struct.unpack_from("<B", blob, pos); pos += 1
        shape = struct.unpack_from("<" + "Q" * nd, blob, pos); pos += 8 * nd
        k, lb, n, total_bits = struct.unpack_from("<IBQQ", blob, pos); pos += 21
        if k == 0:
            data = np.frombuffer(blob, dtype=np.float32, count=n, offset=pos); pos += 4 * n
            out[name] = torch.from_numpy(data.copy()).to(device).reshape(shape)
            continue
        centers = np.frombuffer(blob, dtype=np.float32, count=k, offset=pos); pos += 4 * k
        lengths = np.frombuffer(blob, dtype=np.uint8, count=k, offset=pos); pos += k
        nchunks = (n + CHUNK - 1) // CHUNK
        chunk_bits = np.frombuffer(blob, dtype=np.uint32, count=nchunks, offset=pos); pos += 4 * nchunks
        nwords = (total_bits + 31) // 32
        words = np.frombuffer(blob, dtype=np.int32, count=nwords, offset=pos); pos += 4 * nwords
        if int(chunk_bits.astype(np.int64).sum()) != total_bits:
            raise ValueError("corrupt chunk table")
        labels = decode_indices(torch.from_numpy(words.copy()).to(device), chunk_bits, n, lengths.copy(), k, lb)
        out[name] = ops.gather(torch.from_numpy(centers.copy()).to(device), labels).reshape(shape)
    return out
