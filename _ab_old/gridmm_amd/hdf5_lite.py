"""Minimal read-only HDF5 reader (pure Python + numpy + zlib) for the reference's feature files.

The reference stores its precomputed observations as HDF5 (`clip_p32.hdf5`, `depth.hdf5`: one dataset per
"<scan>_<viewpoint>" key in the root group, numeric arrays, gzip-compressed chunks; preprocess/get_map_feature.py:134,181,
r2r/env.py:80-113) and reads them with h5py, which this image does not have.  The converter of feature_store.py needs
exactly: list the root group, read a whole numeric dataset.  This module does that and nothing else, following the HDF5
File Format Specification v2/v3 for the structures h5py / libhdf5 1.8-1.12 write with default settings:

  superblock v0 / v1  ->  root group symbol-table entry  ->  group B-tree (v1, node type 0) + local heap  ->
  symbol-table nodes (SNOD)  ->  object headers (v1, with continuation blocks)  ->  messages:
    0x0001 dataspace (v1 / v2)      0x0003 datatype (class 0 fixed-point, class 1 floating-point)
    0x0008 data layout v1-v3 (compact | contiguous | chunked: chunk B-tree v1, node type 1)      0x000B filter pipeline (v1 / v2):
    deflate (1), shuffle (2), fletcher32 (3, checksum stripped)
Anything else (new-style groups, v2 object headers, compound / string / variable-length types, external links) raises
Hdf5Unsupported; `feature_store.convert_reference_files` falls back to h5py when it is importable.

Pinned by tests/test_hdf5_lite.py against real HDF5 files written by libhdf5 (tests/golden/hdf5/, from the PyTables test
suite, whose own tests document their contents).
"""
import struct
import zlib

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF


class Hdf5Unsupported(ValueError):
    pass


class _Buf:
    def __init__(self, data):
        self.d = data

    def u(self, off, n):
        return int.from_bytes(self.d[off:off + n], "little")


class Dataset:
    def __init__(self, f, name, shape, dtype, layout, filters):
        self._f, self.name, self.shape, self.dtype, self._layout, self._filters = f, name, tuple(shape), dtype, layout, filters

    def __getitem__(self, idx):
        return self.read()[idx]

    def read(self):
        """The whole dataset as a numpy array in native byte order."""
        f, kind = self._f, self._layout[0]
        n = int(np.prod(self.shape, dtype=np.int64)) if self.shape else 1
        if kind == "contiguous":
            addr, size = self._layout[1], self._layout[2]
            if addr == UNDEF:                      # never written: fill value 0
                return np.zeros(self.shape, self.dtype.newbyteorder("="))
            raw = f._b.d[addr:addr + n * self.dtype.itemsize]
            return np.frombuffer(raw, self.dtype, n).reshape(self.shape).astype(self.dtype.newbyteorder("="))
        if kind == "compact":
            return np.frombuffer(self._layout[1], self.dtype, n).reshape(self.shape).astype(self.dtype.newbyteorder("="))
        btree, chunk = self._layout[1], self._layout[2]       # chunk: dims incl. the element size as last entry
        out = np.zeros(self.shape, self.dtype.newbyteorder("="))
        if btree == UNDEF:
            return out
        cshape = chunk[:-1]
        for offs, size, mask, addr in f._chunks(btree, len(self.shape)):
            raw = bytes(f._b.d[addr:addr + size])
            for k in range(len(self._filters) - 1, -1, -1):     # filters are undone in reverse order
                fid, cd = self._filters[k]
                if mask & (1 << k):
                    continue
                if fid == 1:
                    raw = zlib.decompress(raw)
                elif fid == 2:
                    es = cd[0] if cd else self.dtype.itemsize
                    a = np.frombuffer(raw, np.uint8)
                    m = len(a) // es
                    raw = a[:m * es].reshape(es, m).T.tobytes() + a[m * es:].tobytes()
                elif fid == 3:
                    raw = raw[:-4]
                else:
                    raise Hdf5Unsupported("filter id %d on dataset %s" % (fid, self.name))
            block = np.frombuffer(raw, self.dtype, int(np.prod(cshape))).reshape(cshape)
            sel_out = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cshape, self.shape))
            sel_in = tuple(slice(0, s.stop - s.start) for s in sel_out)
            out[sel_out] = block[sel_in]
        return out


class File:
    """f = File(path); f.keys(); key in f; f[key][...]   (root-group datasets only, like the reference's files)."""

    def __init__(self, path, mode="r"):
        if mode != "r":
            raise Hdf5Unsupported("read-only")
        self._b = _Buf(np.memmap(path, np.uint8, "r"))
        b = self._b
        base = None
        for off in (0, 512, 1024, 2048):
            if bytes(b.d[off:off + 8]) == b"\x89HDF\r\n\x1a\n":
                base = off
                break
        if base is None:
            raise ValueError("%s: not an HDF5 file" % path)
        ver = b.u(base + 8, 1)
        if ver not in (0, 1):
            raise Hdf5Unsupported("superblock version %d (files written with libver='latest' are not supported)" % ver)
        self._so, self._sl = b.u(base + 13, 1), b.u(base + 14, 1)
        if (self._so, self._sl) != (8, 8):
            raise Hdf5Unsupported("offset / length size %d / %d" % (self._so, self._sl))
        p = base + 24 + (4 if ver == 1 else 0)
        p += 4 * 8                                       # base address, free-space address, end of file, driver info
        # root group symbol table entry: link name offset, object header address, cache type, reserved, scratch
        self._root_header = b.u(p + 8, 8)
        cache_type = b.u(p + 16, 4)
        if cache_type == 1:
            self._root_btree, self._root_heap = b.u(p + 24, 8), b.u(p + 32, 8)
        else:
            msgs = self._messages(self._root_header)
            st = [m for m in msgs if m[0] == 0x11]
            if not st:
                raise Hdf5Unsupported("root group without a symbol table (new-style group)")
            self._root_btree, self._root_heap = self._b.u(st[0][1], 8), self._b.u(st[0][1] + 8, 8)
        self._links = None

    # ---- groups (old style: B-tree v1 of symbol-table nodes + local heap of names)
    def _heap_data(self, addr):
        b = self._b
        if bytes(b.d[addr:addr + 4]) != b"HEAP":
            raise ValueError("local heap signature")
        return b.u(addr + 24, 8)

    def _walk_group(self, node, heap_data, out):
        b = self._b
        sig = bytes(b.d[node:node + 4])
        if sig == b"TREE":
            if b.u(node + 4, 1) != 0:
                raise ValueError("group B-tree node type")
            n = b.u(node + 6, 2)
            p = node + 8 + 16                            # skip the sibling addresses
            for i in range(n):
                child = b.u(p + 8 + i * 16, 8)           # key (8) | child (8) | key | child ...
                self._walk_group(child, heap_data, out)
        elif sig == b"SNOD":
            n = b.u(node + 6, 2)
            p = node + 8
            for i in range(n):
                e = p + i * 40
                name_off, header = b.u(e, 8), b.u(e + 8, 8)
                s = heap_data + name_off
                end = s
                while b.d[end] != 0:
                    end += 1
                out[bytes(b.d[s:end]).decode("utf-8")] = header
        else:
            raise ValueError("unexpected group node signature %r" % sig)

    def _load_links(self):
        if self._links is None:
            out = {}
            self._walk_group(self._root_btree, self._heap_data(self._root_heap), out)
            self._links = out
        return self._links

    def keys(self):
        return list(self._load_links().keys())

    def __contains__(self, key):
        return key in self._load_links()

    def __len__(self):
        return len(self._load_links())

    # ---- object headers (v1)
    def _messages(self, addr):
        """[(type, data offset, size)] of a version-1 object header incl. its continuation blocks."""
        b = self._b
        if bytes(b.d[addr:addr + 4]) == b"OHDR":
            raise Hdf5Unsupported("version-2 object header")
        if b.u(addr, 1) != 1:
            raise ValueError("object header version %d" % b.u(addr, 1))
        n_msgs, size = b.u(addr + 2, 2), b.u(addr + 8, 4)
        blocks, msgs = [(addr + 16, size)], []
        while blocks and len(msgs) < n_msgs:
            p, left = blocks.pop(0)
            while left >= 8 and len(msgs) < n_msgs:
                t, s = b.u(p, 2), b.u(p + 2, 2)
                data = p + 8
                if t == 0x10:                            # continuation: offset, length
                    blocks.append((b.u(data, 8), b.u(data + 8, 8)))
                msgs.append((t, data, s))
                p += 8 + s
                left -= 8 + s
        return msgs

    def _datatype(self, p):
        b = self._b
        cls_ver = b.u(p, 1)
        cls, bits0 = cls_ver & 0x0F, b.u(p + 1, 1)
        size = b.u(p + 4, 4)
        order = ">" if (bits0 & 1) else "<"
        if cls == 0:
            signed = (bits0 >> 3) & 1
            return np.dtype("%s%s%d" % (order, "i" if signed else "u", size))
        if cls == 1:
            if size not in (2, 4, 8):
                raise Hdf5Unsupported("float of %d bytes" % size)
            return np.dtype("%sf%d" % (order, size))
        raise Hdf5Unsupported("datatype class %d (only integers and IEEE floats)" % cls)

    def _dataspace(self, p):
        b = self._b
        ver, rank, flags = b.u(p, 1), b.u(p + 1, 1), b.u(p + 2, 1)
        q = p + (8 if ver == 1 else 4)
        return [b.u(q + 8 * i, 8) for i in range(rank)]

    def _filters(self, p):
        b = self._b
        ver, n = b.u(p, 1), b.u(p + 1, 1)
        q = p + (8 if ver == 1 else 2)
        out = []
        for _ in range(n):
            fid = b.u(q, 2)
            if ver == 1 or fid >= 256:
                name_len = b.u(q + 2, 2)
                flags, ncd = b.u(q + 4, 2), b.u(q + 6, 2)
                q += 8 + (name_len + 7) // 8 * 8 if ver == 1 else 8 + name_len
            else:
                flags, ncd = b.u(q + 2, 2), b.u(q + 4, 2)
                q += 6
            cd = [b.u(q + 4 * i, 4) for i in range(ncd)]
            q += 4 * ncd
            if ver == 1 and ncd % 2:
                q += 4
            out.append((fid, cd))
        return out

    def _layout(self, p):
        b = self._b
        ver = b.u(p, 1)
        if ver in (1, 2):                                # libhdf5 < 1.6.3: rank | class | 5 reserved | address | dims (u32)
            rank, cls = b.u(p + 1, 1), b.u(p + 2, 1)
            q = p + 8
            addr = UNDEF
            if cls != 0:
                addr, q = b.u(q, 8), q + 8
            dims = [b.u(q + 4 * i, 4) for i in range(rank)]
            q += 4 * rank
            if cls == 0:
                n = b.u(q, 4)
                return ("compact", bytes(b.d[q + 4:q + 4 + n]))
            if cls == 1:
                return ("contiguous", addr, None)
            if cls == 2:                                 # the chunk dims already carry the element size as last entry
                return ("chunked", addr, dims)
            raise Hdf5Unsupported("layout class %d" % cls)
        if ver != 3:
            raise Hdf5Unsupported("data layout message version %d" % ver)
        cls = b.u(p + 1, 1)
        if cls == 0:
            n = b.u(p + 2, 2)
            return ("compact", bytes(b.d[p + 4:p + 4 + n]))
        if cls == 1:
            return ("contiguous", b.u(p + 2, 8), b.u(p + 10, 8))
        if cls == 2:
            rank = b.u(p + 2, 1)
            return ("chunked", b.u(p + 3, 8), [b.u(p + 11 + 4 * i, 4) for i in range(rank)])
        raise Hdf5Unsupported("layout class %d" % cls)

    def _chunks(self, node, rank):
        """Leaves of a chunk B-tree (v1, node type 1): (offsets, stored size, filter mask, address)."""
        b = self._b
        if bytes(b.d[node:node + 4]) != b"TREE" or b.u(node + 4, 1) != 1:
            raise ValueError("chunk B-tree node")
        level, n = b.u(node + 5, 1), b.u(node + 6, 2)
        key_size = 8 + 8 * (rank + 1)
        p = node + 8 + 16
        for i in range(n):
            k = p + i * (key_size + 8)
            size, mask = b.u(k, 4), b.u(k + 4, 4)
            offs = [b.u(k + 8 + 8 * d, 8) for d in range(rank)]
            child = b.u(k + key_size, 8)
            if level == 0:
                yield offs, size, mask, child
            else:
                yield from self._chunks(child, rank)

    def __getitem__(self, key):
        links = self._load_links()
        if key not in links:
            raise KeyError(key)
        shape = dtype = layout = None
        filters = []
        for t, data, size in self._messages(links[key]):
            if t == 0x01:
                shape = self._dataspace(data)
            elif t == 0x03:
                dtype = self._datatype(data)
            elif t == 0x08:
                layout = self._layout(data)
            elif t == 0x0B:
                filters = self._filters(data)
        if shape is None or dtype is None or layout is None:
            raise Hdf5Unsupported("%s is not a plain dataset (group, or a message this reader does not know)" % key)
        return Dataset(self, key, shape, dtype, layout, filters)
