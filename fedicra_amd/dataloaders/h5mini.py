"""A read-only decoder for the subset of HDF5 the reference's data sets use (the files of data/FAZ_h5, data/ODOC_h5,
data/Polyp_h5 that /root/reference/code/dataloaders/dataset.py:84-96 opens with ``h5py.File(path, 'r')`` and reads with
``h5f['image'][:]``), for boxes without h5py.  Host-side IO: a split is decoded once and then lives in HBM
(``BaseDataSets.resident``), so this is not on the timed path.

What those files are (all 3115 of them: superblock version 0, 8-byte offsets and lengths): a root group whose links sit
in a version-1 group B-tree + local heap + symbol-table nodes; each data set has a version-1 object header with a
simple dataspace, a fixed-point or IEEE floating-point datatype, a version-3 layout message (chunked, or contiguous /
compact) and a version-1 filter pipeline of deflate (h5py's ``compression='gzip'``), optionally preceded by shuffle;
chunks are indexed by a version-1 chunk B-tree.  Everything else (newer superblocks, fractal-heap groups, variable
length / compound types, other filters, external storage) raises ``H5Error`` -- never a guess.

The structures follow the published HDF5 File Format Specification (version 1.1 structures of the "III. Disk Format"
chapters: superblock, B-link trees, symbol table entries, local heaps, object header messages 0x0001 / 0x0003 /
0x0008 / 0x000B / 0x0010 / 0x0011).  Usage mirrors the two h5py calls the reference makes::

    with File(path, "r") as h5f:
        image = h5f["image"][:]
"""
from __future__ import annotations

import zlib

import numpy as np

_SIG = b"\x89HDF\r\n\x1a\n"
_UNDEF = (1 << 64) - 1


class H5Error(ValueError):
    """The file uses an HDF5 feature outside the subset above, or is damaged."""


_MAX_NODES = 1 << 20          # B-tree / continuation blocks visited per object: far above any real file, bounds a damaged one


def _u(b, o, n):
    if o < 0 or o + n > len(b):
        raise H5Error("read of {} bytes at {} past the end of the file ({} bytes)".format(n, o, len(b)))
    return int.from_bytes(b[o:o + n], "little")


def _span(b, o, n):
    """b[o:o+n], bounds-checked (a plain slice silently shortens, a plain index raises IndexError)."""
    if o < 0 or n < 0 or o + n > len(b):
        raise H5Error("read of {} bytes at {} past the end of the file ({} bytes)".format(n, o, len(b)))
    return b[o:o + n]


def _guard(fn):
    """Whatever a damaged file makes the decoder trip over surfaces as H5Error -- the promise of the module docstring."""
    import functools

    @functools.wraps(fn)
    def wrapped(*a, **k):
        try:
            return fn(*a, **k)
        except H5Error:
            raise
        except (IndexError, ValueError, OverflowError, MemoryError, zlib.error, UnicodeDecodeError) as e:
            raise H5Error("damaged or unsupported HDF5 structure: {}: {}".format(type(e).__name__, e)) from e
    return wrapped


class Dataset:
    """One data set: ``shape``, ``dtype`` and numpy-style reads of the whole array (``ds[:]``, ``ds[()]``, ``ds[...]``;
    any other index is applied to the decoded array)."""

    @_guard
    def __init__(self, f, name, addr):
        self._f, self.name = f, name
        self.shape = self.dtype = self._layout = None
        self._filters = []
        self._cache = None
        for mtype, m in f._messages(addr):
            if mtype == 0x0001:
                self.shape = self._dataspace(m)
            elif mtype == 0x0003:
                self.dtype = self._datatype(m)
            elif mtype == 0x0008:
                self._layout = self._layout_msg(m)
            elif mtype == 0x000B:
                self._filters = self._pipeline(m)
        if self.shape is None or self.dtype is None or self._layout is None:
            raise H5Error("'{}' is not a data set (dataspace / datatype / layout message missing)".format(name))

    @staticmethod
    def _dataspace(m):
        ver, rank = m[0], m[1]
        if ver == 1:
            o = 8
        elif ver == 2:
            if m[3] == 2:
                raise H5Error("null dataspace")
            o = 4
        else:
            raise H5Error("dataspace message version {}".format(ver))
        return tuple(_u(m, o + 8 * i, 8) for i in range(rank))

    @staticmethod
    def _datatype(m):
        cls, ver, bits0, size = m[0] & 0x0F, m[0] >> 4, m[1], _u(m, 4, 4)
        if ver not in (1, 2, 3):
            raise H5Error("datatype message version {}".format(ver))
        order = ">" if bits0 & 1 else "<"
        if cls == 0 and size in (1, 2, 4, 8):                  # fixed point; bit 3 = signed
            prec_off, prec = _u(m, 8, 2), _u(m, 10, 2)
            if prec_off != 0 or prec != 8 * size:
                raise H5Error("fixed-point type with padding bits")
            return np.dtype(order + ("i" if bits0 & 8 else "u") + str(size))
        if cls == 1 and size in (2, 4, 8):                     # IEEE floating point
            if bits0 & 0x40:
                raise H5Error("VAX-endian floating point")
            return np.dtype(order + "f" + str(size))
        raise H5Error("datatype class {} of {} bytes".format(cls, size))

    @staticmethod
    def _layout_msg(m):
        if m[0] != 3:
            raise H5Error("data layout message version {}".format(m[0]))
        if m[1] == 0:                                            # compact: the data sit in the message
            n = _u(m, 2, 2)
            return ("compact", bytes(m[4:4 + n]))
        if m[1] == 1:
            return ("contiguous", _u(m, 2, 8), _u(m, 10, 8))
        if m[1] == 2:
            nd = m[2]                                            # rank + 1: the last "dimension" is the element size
            return ("chunked", _u(m, 3, 8), tuple(_u(m, 11 + 4 * i, 4) for i in range(nd)))
        raise H5Error("data layout class {}".format(m[1]))

    @staticmethod
    def _pipeline(m):
        if m[0] != 1:
            raise H5Error("filter pipeline message version {}".format(m[0]))
        out, p = [], 8
        for _ in range(m[1]):
            fid, nlen, ncv = _u(m, p, 2), _u(m, p + 2, 2), _u(m, p + 6, 2)
            p += 8 + (nlen + 7) // 8 * 8                         # the name is padded to a multiple of 8
            p += 4 * (ncv + (ncv & 1))                           # client values, padded to an even count
            if fid not in (1, 2):
                raise H5Error("filter {} (only deflate = 1 and shuffle = 2 are decoded)".format(fid))
            out.append(fid)
        return out

    def _unfilter(self, raw, mask):
        for k in range(len(self._filters) - 1, -1, -1):          # decode in the reverse of the write order
            if mask >> k & 1:                                    # the writer skipped this filter for this chunk
                continue
            if self._filters[k] == 1:
                raw = zlib.decompress(raw)
            else:                                                # shuffle: byte planes back to elements
                es = self.dtype.itemsize
                a = np.frombuffer(raw, np.uint8)
                nel = len(a) // es
                raw = a[:nel * es].reshape(es, nel).T.tobytes() + a[nel * es:].tobytes()
        return raw

    @_guard
    def _read(self):
        b, shape, dtype = self._f._b, self.shape, self.dtype
        count = int(np.prod(shape, dtype=np.int64)) if shape else 1
        kind = self._layout[0]
        if kind == "compact":
            return np.frombuffer(self._layout[1], dtype, count).reshape(shape).copy()
        if kind == "contiguous":
            addr = self._layout[1]
            if addr == _UNDEF:                                   # never written: the fill value (0 without a fill message)
                return np.zeros(shape, dtype)
            if addr + count * dtype.itemsize > len(b):
                raise H5Error("contiguous data of '{}' run past the end of the file".format(self.name))
            return np.frombuffer(b, dtype, count, addr).reshape(shape).copy()
        root, cdims = self._layout[1], self._layout[2]
        rank = len(shape)
        if len(cdims) != rank + 1 or cdims[rank] != dtype.itemsize:
            raise H5Error("chunk dimensions {} do not match a rank-{} {} data set".format(cdims, rank, dtype))
        cshape = cdims[:rank]
        ccount = int(np.prod(cshape, dtype=np.int64))
        out = np.zeros(shape, dtype)
        if root == _UNDEF:
            return out
        key = 8 + 8 * (rank + 1)                                 # chunk size, filter mask, rank+1 offsets
        todo, seen = [root], set()
        while todo:
            addr = todo.pop()
            if addr in seen or len(seen) >= _MAX_NODES:          # a cyclic (damaged) tree must not loop forever
                raise H5Error("chunk B-tree of '{}' revisits node {} (cyclic or oversized)".format(self.name, addr))
            seen.add(addr)
            if _span(b, addr, 4) != b"TREE" or _u(b, addr + 4, 1) != 1:
                raise H5Error("chunk B-tree node expected at {}".format(addr))
            level, n = _u(b, addr + 5, 1), _u(b, addr + 6, 2)
            p = addr + 24
            for _ in range(n):
                csize, mask = _u(b, p, 4), _u(b, p + 4, 4)
                offs = tuple(_u(b, p + 8 + 8 * j, 8) for j in range(rank))
                child = _u(b, p + key, 8)
                p += key + 8
                if level:
                    todo.append(child)
                    continue
                if child + csize > len(b):
                    raise H5Error("chunk of '{}' runs past the end of the file".format(self.name))
                raw = self._unfilter(_span(b, child, csize), mask)
                if len(raw) < ccount * dtype.itemsize:
                    raise H5Error("chunk of '{}' decodes to {} bytes, {} expected".format(
                        self.name, len(raw), ccount * dtype.itemsize))
                chunk = np.frombuffer(raw, dtype, ccount).reshape(cshape)
                dst = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cshape, shape))
                out[dst] = chunk[tuple(slice(0, d.stop - d.start) for d in dst)]     # edge chunks are stored whole
        return out

    def __getitem__(self, index):
        if self._cache is None:                                  # decoded once per Dataset object (h5py hands out fresh arrays:
            self._cache = self._read()                           # so does this -- copies of the cached decode)
        a = self._cache.copy()
        whole = index is Ellipsis or (isinstance(index, tuple) and not index) or \
            (isinstance(index, slice) and index == slice(None))
        if whole:
            return a
        return a[index]

    def __len__(self):
        return self.shape[0]


class File:
    """``File(path, 'r')``: the root group's data sets by name (``f['image']``, ``'mask' in f``, ``f.keys()``)."""

    @_guard
    def __init__(self, path, mode="r"):
        if mode != "r":
            raise H5Error("h5mini only reads (mode 'r'), got mode {!r}".format(mode))
        with open(path, "rb") as fh:
            self._b = b = fh.read()
        self.filename = path
        if b[:8] != _SIG:
            raise H5Error("{}: no HDF5 signature at offset 0".format(path))
        if _u(b, 8, 1) != 0:
            raise H5Error("{}: superblock version {} (only version 0 is decoded)".format(path, b[8]))
        if (_u(b, 13, 1), _u(b, 14, 1)) != (8, 8):
            raise H5Error("{}: {}-byte offsets / {}-byte lengths (only 8 / 8)".format(path, b[13], b[14]))
        if _u(b, 24, 8) != 0:
            raise H5Error("{}: non-zero base address".format(path))
        self._links = self._group(56)                            # root symbol-table entry: 24 + 4 addresses
        self._open = {}

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def close(self):
        self._b, self._open = b"", {}

    def _entry(self, o):
        b = self._b
        e = {"name": _u(b, o, 8), "header": _u(b, o + 8, 8)}
        if _u(b, o + 16, 4) == 1:                                # cached group: B-tree and heap in the scratch pad
            e["btree"], e["heap"] = _u(b, o + 24, 8), _u(b, o + 32, 8)
        return e

    def _group(self, entry_at):
        """name -> object header address of every link of the group whose symbol-table entry is at ``entry_at``."""
        b = self._b
        e = self._entry(entry_at)
        if "btree" not in e:
            for mtype, m in self._messages(e["header"]):
                if mtype == 0x0011:
                    e["btree"], e["heap"] = _u(m, 0, 8), _u(m, 8, 8)
        if "btree" not in e:
            raise H5Error("{}: the root group has no symbol table (new-style groups are not decoded)".format(self.filename))
        if _span(b, e["heap"], 4) != b"HEAP":
            raise H5Error("local heap expected at {}".format(e["heap"]))
        names = _u(b, e["heap"] + 24, 8)                         # address of the heap's data segment
        links, todo, seen = {}, [e["btree"]], set()
        while todo:
            addr = todo.pop()
            if addr in seen or len(seen) >= _MAX_NODES:          # a cyclic (damaged) tree must not loop forever
                raise H5Error("group B-tree revisits node {} (cyclic or oversized)".format(addr))
            seen.add(addr)
            if _span(b, addr, 4) == b"SNOD":
                for i in range(_u(b, addr + 6, 2)):
                    s = self._entry(addr + 8 + 40 * i)
                    start = names + s["name"]
                    end = b.find(b"\0", start) if 0 <= start < len(b) else -1
                    if end < 0:
                        raise H5Error("link name at {} is not terminated".format(start))
                    links[b[start:end].decode("utf-8")] = s["header"]
                continue
            if _span(b, addr, 4) != b"TREE" or _u(b, addr + 4, 1) != 0:
                raise H5Error("group B-tree node expected at {}".format(addr))
            level, n = _u(b, addr + 5, 1), _u(b, addr + 6, 2)
            for i in range(n):                                   # key0 child0 key1 child1 ... : children at odd slots
                child = _u(b, addr + 24 + 16 * i + 8, 8)
                todo.append(child)                               # level 0 children are symbol-table nodes
            del level
        return links

    def _messages(self, addr):
        """(type, body) of every message of a version-1 object header, following continuation blocks."""
        b = self._b
        if _span(b, addr, 4) == b"OHDR":
            raise H5Error("version-2 object header at {} (only version 1 is decoded)".format(addr))
        if _u(b, addr, 1) != 1:
            raise H5Error("object header version {} at {}".format(b[addr], addr))
        total, out = _u(b, addr + 2, 2), []
        blocks, seen = [(addr + 16, _u(b, addr + 8, 4))], set()
        while blocks and len(out) < total:
            p, size = blocks.pop(0)
            if p in seen or len(seen) >= _MAX_NODES:             # continuation blocks that point back at each other
                raise H5Error("object header at {} revisits continuation block {}".format(addr, p))
            seen.add(p)
            end = p + size
            while p + 8 <= end and len(out) < total:
                mtype, msize = _u(b, p, 2), _u(b, p + 2, 2)
                body = _span(b, p + 8, msize)
                if mtype == 0x0010:
                    blocks.append((_u(body, 0, 8), _u(body, 8, 8)))
                out.append((mtype, body))
                p += 8 + msize
        return out

    def keys(self):
        return list(self._links)

    def __contains__(self, name):
        return name in self._links

    def __iter__(self):
        return iter(self._links)

    def __len__(self):
        return len(self._links)

    @_guard
    def __getitem__(self, name):
        if name not in self._links:
            raise KeyError("Unable to open object (object '{}' doesn't exist)".format(name))
        if name not in self._open:
            self._open[name] = Dataset(self, name, self._links[name])
        return self._open[name]
