"""A minimal reader for the HDF5 files the reference's icicle test holds (models/kinematic_2D/tests/paper_GMD_2015/fig_a/refdata): superblock
version 0, old-style groups (symbol table: B-tree v1 + local heap), version-1 object headers, contiguous or chunked (B-tree v1)
datasets of little-endian floats / integers, optionally deflate- and shuffle-filtered.  Standard library + numpy only (the image has no
h5py).  Not a general HDF5 implementation: it raises on anything else.

    import h5min; f = h5min.File(path); f.keys(); f["th"]  -> numpy array
"""
import struct
import zlib

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF


class File:
    def __init__(self, path):
        self.b = open(path, "rb").read()
        b = self.b
        if b[:8] != b"\x89HDF\r\n\x1a\n" or b[8] != 0:
            raise ValueError("not an HDF5 file with a version-0 superblock")
        self.so, self.sl = b[13], b[14]                       # size of offsets / lengths
        if (self.so, self.sl) != (8, 8):
            raise ValueError("offsets / lengths are not 8 bytes")
        # superblock v0: ... base address at 24, then free-space, eof, driver; root symbol table entry at 24 + 4*8
        root = 24 + 4 * 8
        self.root_objhdr = self._u64(root + 8)
        cache_type = struct.unpack_from("<I", b, root + 16)[0]
        if cache_type != 1:
            raise ValueError("root group without cached symbol table")
        self.datasets = {}
        self._walk_group(self._u64(root + 24), self._u64(root + 32), "")

    def _u64(self, off):
        return struct.unpack_from("<Q", self.b, off)[0]

    # ---- groups
    def _heap_data(self, heap_addr):
        b = self.b
        if b[heap_addr:heap_addr + 4] != b"HEAP":
            raise ValueError("local heap expected")
        return self._u64(heap_addr + 24)                      # address of the data segment

    def _walk_group(self, btree, heap, prefix):
        hd = self._heap_data(heap)
        for name_off, objhdr, cache_type, scratch in self._btree_group_entries(btree):
            e = self.b.index(b"\x00", hd + name_off)
            name = self.b[hd + name_off:e].decode()
            if cache_type == 1:                               # a sub-group with cached B-tree / heap addresses
                self._walk_group(struct.unpack_from("<Q", scratch, 0)[0], struct.unpack_from("<Q", scratch, 8)[0], prefix + name + "/")
            else:
                self.datasets[prefix + name] = objhdr

    def _btree_group_entries(self, addr):
        b = self.b
        if b[addr:addr + 4] != b"TREE" or b[addr + 4] != 0:
            raise ValueError("group B-tree node expected")
        level, used = b[addr + 5], struct.unpack_from("<H", b, addr + 6)[0]
        p = addr + 8 + 16                                      # past the sibling addresses
        p += 8                                                 # key 0
        for _ in range(used):
            child = self._u64(p)
            p += 16                                            # child address + next key
            if level > 0:
                yield from self._btree_group_entries(child)
            else:
                yield from self._snod(child)

    def _snod(self, addr):
        b = self.b
        if b[addr:addr + 4] != b"SNOD":
            raise ValueError("symbol table node expected")
        n = struct.unpack_from("<H", b, addr + 6)[0]
        for i in range(n):
            e = addr + 8 + i * 40
            yield self._u64(e), self._u64(e + 8), struct.unpack_from("<I", b, e + 16)[0], b[e + 24:e + 40]

    # ---- object headers (version 1)
    def _messages(self, addr):
        b = self.b
        if b[addr] != 1:
            raise ValueError("object header version %d not supported" % b[addr])
        n_msg = struct.unpack_from("<H", b, addr + 2)[0]
        size = struct.unpack_from("<I", b, addr + 8)[0]
        blocks = [(addr + 16, size)]
        out = []
        while blocks and len(out) < n_msg:
            p, left = blocks.pop(0)
            end = p + left
            while p + 8 <= end and len(out) < n_msg:
                mtype, msize = struct.unpack_from("<HH", b, p)
                body = p + 8
                if mtype == 0x10:                              # continuation
                    blocks.append((self._u64(body), self._u64(body + 8)))
                else:
                    out.append((mtype, body, msize))
                p = body + msize
        return out

    def keys(self):
        return sorted(self.datasets)

    def __getitem__(self, name):
        b = self.b
        shape = dtype = layout = None
        filters = []
        for mtype, p, sz in self._messages(self.datasets[name]):
            if mtype == 0x01:                                  # dataspace
                ver, rank, flags = b[p], b[p + 1], b[p + 2]
                q = p + (8 if ver == 1 else 4)
                shape = tuple(self._u64(q + 8 * i) for i in range(rank))
            elif mtype == 0x03:                                # datatype
                cls, size = b[p] & 0x0F, struct.unpack_from("<I", b, p + 4)[0]
                if b[p + 1] & 1:
                    raise ValueError("big-endian data")
                if cls == 1:
                    dtype = np.dtype("<f%d" % size)
                elif cls == 0:
                    dtype = np.dtype("<%s%d" % ("i" if b[p + 1] & 8 else "u", size))
                else:
                    raise ValueError("datatype class %d not supported" % cls)
            elif mtype == 0x08:                                # data layout
                if b[p] != 3:
                    raise ValueError("data layout version %d not supported" % b[p])
                lclass = b[p + 1]
                if lclass == 1:
                    layout = ("contiguous", self._u64(p + 2), self._u64(p + 10))
                elif lclass == 2:
                    rank = b[p + 2]
                    layout = ("chunked", self._u64(p + 3), tuple(struct.unpack_from("<I", b, p + 11 + 4 * i)[0] for i in range(rank)))
                elif lclass == 0:
                    layout = ("compact", p + 4, struct.unpack_from("<H", b, p + 2)[0])
            elif mtype == 0x0B:                                # filter pipeline
                ver, nf = b[p], b[p + 1]
                q = p + (8 if ver == 1 else 2)
                for _ in range(nf):
                    fid, nlen, flags, ncd = struct.unpack_from("<HHHH", b, q)
                    q += 8
                    if ver == 1 or fid >= 256:
                        q += (nlen + 7) // 8 * 8 if ver == 1 else nlen
                    q += 4 * ncd
                    if ver == 1 and ncd % 2:
                        q += 4
                    filters.append(fid)
        if shape is None or dtype is None or layout is None:
            raise ValueError("dataset %s: incomplete header" % name)
        n = int(np.prod(shape)) if shape else 1
        if layout[0] == "contiguous":
            if layout[1] == UNDEF:
                return np.zeros(shape, dtype)
            return np.frombuffer(b, dtype, n, layout[1]).reshape(shape).copy()
        if layout[0] == "compact":
            return np.frombuffer(b, dtype, n, layout[1]).reshape(shape).copy()
        out = np.zeros(shape, dtype)
        chunk = layout[2][:-1]                                 # (the last entry is the element size)
        for offs, addr, nbytes, mask in self._chunks(layout[1], len(chunk)):
            raw = b[addr:addr + nbytes]
            for fid in reversed(filters):
                if fid == 1:
                    raw = zlib.decompress(raw)
                elif fid == 2:
                    a = np.frombuffer(raw, np.uint8).reshape(dtype.itemsize, -1)
                    raw = a.T.tobytes()
                else:
                    raise ValueError("filter %d not supported" % fid)
            c = np.frombuffer(raw, dtype).reshape(chunk)
            sl = tuple(slice(o, min(o + s, e)) for o, s, e in zip(offs, chunk, shape))
            out[sl] = c[tuple(slice(0, s.stop - s.start) for s in sl)]
        return out

    def _chunks(self, addr, rank):
        b = self.b
        if addr == UNDEF:
            return
        if b[addr:addr + 4] != b"TREE" or b[addr + 4] != 1:
            raise ValueError("chunk B-tree node expected")
        level, used = b[addr + 5], struct.unpack_from("<H", b, addr + 6)[0]
        p = addr + 8 + 16
        ksz = 8 + 8 * (rank + 1)
        for _ in range(used):
            nbytes, mask = struct.unpack_from("<II", b, p)
            offs = tuple(self._u64(p + 8 + 8 * i) for i in range(rank))
            child = self._u64(p + ksz)
            p += ksz + 8
            if level > 0:
                yield from self._chunks(child, rank)
            else:
                yield offs, child, nbytes, mask
