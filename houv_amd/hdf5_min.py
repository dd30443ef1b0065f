"""Dependency-free HDF5 subset for the files at the edge of the registration path (h5py is not always installed):

* ``write_h5(path, {name: array})`` -- what ``test.py:70-71`` needs (``results.h5`` with dataset ``results`` f32 [N,4,4]):
  superblock v0, one root group (symbol table), contiguous little-endian integer / float datasets, the layout libhdf5
  itself emits with ``libver='earliest'``.
* ``H5File(path)`` -- what ``dataset.py:189-238, 354-372`` needs (``MVP_*_RG.h5``: ``f['src']``, ``np.array(f[k])``,
  ``f[k][l:r]``): superblock v0-v3, object headers v1/v2, old-style groups (symbol table + B-tree v1 + local heap) and
  compact new-style groups (link messages), contiguous / compact / chunked (B-tree v1) layouts, deflate, shuffle and
  fletcher32 filters, fixed-point and IEEE float types of either byte order.

Chunk indexes of the v4 layout (``libver='latest'``): single-chunk, implicit, fixed array, and -- for resizable datasets --
extensible array (one unlimited dimension) and v2 B-tree (several).  Not supported (raises ``H5FormatError``): strings /
compounds / references, external or virtual storage.  Format: "HDF5 File Format Specification
Version 3.0".  tests/test_hdf5_min.py checks both directions against fixtures produced with libhdf5 1.10.6 and, where
the image has them, against ``h5dump`` / ``libhdf5.so`` themselves."""
import mmap
import struct
import zlib

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF
SIGNATURE = b"\x89HDF\r\n\x1a\n"


class H5FormatError(RuntimeError):
    pass


# ---------------------------------------------------------------------------------------------------------------------
# reader
# ---------------------------------------------------------------------------------------------------------------------
def _parse_datatype(b):
    """Datatype message -> numpy dtype (classes 0 fixed-point and 1 floating-point only)."""
    cls, ver = b[0] & 0x0F, b[0] >> 4
    bits0 = b[1]
    size = int.from_bytes(b[4:8], "little")
    order = ">" if (bits0 & 1) else "<"
    if ver not in (1, 2, 3):
        raise H5FormatError("datatype message version %d" % ver)
    if cls == 0:
        signed = bool(bits0 & 0x08)
        prec = int.from_bytes(b[10:12], "little")
        if prec != size * 8 or size not in (1, 2, 4, 8):
            raise H5FormatError("fixed-point type with padding bits (size %d, precision %d)" % (size, prec))
        return np.dtype("%s%s%d" % (order, "i" if signed else "u", size))
    if cls == 1:
        prec = int.from_bytes(b[10:12], "little")
        eloc, esize, mloc, msize = b[12], b[13], b[14], b[15]
        ieee = {2: (10, 5, 0, 10), 4: (23, 8, 0, 23), 8: (52, 11, 0, 52)}
        if size not in ieee or prec != size * 8 or (eloc, esize, mloc, msize) != ieee[size]:
            raise H5FormatError("non-IEEE floating-point type")
        if b[1] & 0x40:
            raise H5FormatError("VAX-endian floating-point type")
        return np.dtype("%sf%d" % (order, size))
    raise H5FormatError("datatype class %d is not supported (only integers and floats)" % cls)


class Dataset:
    """Read-only dataset: ``shape``, ``dtype``, ``ds[...]`` / ``ds[l:r]`` / ``np.array(ds)`` like h5py's."""

    def __init__(self, f, name, msgs):
        self._f, self.name = f, name
        self.shape = self.dtype = None
        self.maxshape = None
        self._layout = None
        self._filters = []
        for typ, data in msgs:
            if typ == 0x0001:
                self.shape, self.maxshape = self._parse_space(data)
            elif typ == 0x0003:
                self.dtype = _parse_datatype(data)
            elif typ == 0x0008:
                self._layout = self._parse_layout(data)
            elif typ == 0x000B:
                self._filters = self._parse_filters(data)
        if self.shape is None or self.dtype is None or self._layout is None:
            raise H5FormatError("%s: not a dataset (dataspace / datatype / layout message missing)" % name)

    # -- messages --
    @staticmethod
    def _parse_space(b):
        ver, rank, flags = b[0], b[1], b[2]
        if ver == 1:
            off = 8
        elif ver == 2:
            if b[3] == 2:
                raise H5FormatError("null dataspace")
            off = 4
        else:
            raise H5FormatError("dataspace message version %d" % ver)
        dims = tuple(int.from_bytes(b[off + 8 * i:off + 8 * i + 8], "little") for i in range(rank))
        maxd = dims
        if flags & 1:         # maximum dimensions present (UNDEF = unlimited)
            off += 8 * rank
            maxd = tuple(int.from_bytes(b[off + 8 * i:off + 8 * i + 8], "little") for i in range(rank))
        return dims, maxd

    def _parse_layout(self, b):
        ver = b[0]
        if ver == 3 or ver == 4:
            cls = b[1]
            if cls == 0:
                n = int.from_bytes(b[2:4], "little")
                return ("compact", bytes(b[4:4 + n]))
            if cls == 1:
                return ("contiguous", int.from_bytes(b[2:10], "little"), int.from_bytes(b[10:18], "little"))
            if cls == 2 and ver == 3:
                nd = b[2]
                addr = int.from_bytes(b[3:11], "little")
                dims = [int.from_bytes(b[11 + 4 * i:15 + 4 * i], "little") for i in range(nd)]
                return ("chunked", addr, tuple(dims[:-1]), dims[-1])
            if cls == 2 and ver == 4:
                flags, nd, enc = b[2], b[3], b[4]
                dims = [int.from_bytes(b[5 + enc * i:5 + enc * (i + 1)], "little") for i in range(nd)]
                p = 5 + enc * nd
                idx = b[p]
                p += 1
                if idx == 1:      # single chunk
                    size = mask = None
                    if flags & 0x02:
                        size = int.from_bytes(b[p:p + 8], "little")
                        mask = int.from_bytes(b[p + 8:p + 12], "little")
                        p += 12
                    return ("single", int.from_bytes(b[p:p + 8], "little"), tuple(dims[:-1]), dims[-1], size, mask)
                if idx == 2:      # implicit: chunks stored back to back, no filters
                    return ("implicit", int.from_bytes(b[p:p + 8], "little"), tuple(dims[:-1]), dims[-1])
                if idx == 3:      # fixed array
                    return ("farray", int.from_bytes(b[p + 1:p + 9], "little"), tuple(dims[:-1]), dims[-1])
                if idx == 4:      # extensible array (one unlimited dimension): 5 creation parameters, then the header address
                    return ("earray", int.from_bytes(b[p + 5:p + 13], "little"), tuple(dims[:-1]), dims[-1])
                if idx == 5:      # v2 B-tree (two or more unlimited dimensions): node size (4), split %, merge %, header address
                    return ("btree2", int.from_bytes(b[p + 6:p + 14], "little"), tuple(dims[:-1]), dims[-1])
                raise H5FormatError("%s: chunk index type %d is not supported" % (self.name, idx))
            raise H5FormatError("%s: layout class %d" % (self.name, cls))
        if ver in (1, 2):     # libhdf5 < 1.6.3: dims follow the address; for chunked storage the last one is the element size
            nd, cls = b[1], b[2]
            p = 8
            addr = None
            if cls != 0:
                addr = int.from_bytes(b[p:p + 8], "little")
                p += 8
            dims = [int.from_bytes(b[p + 4 * i:p + 4 * i + 4], "little") for i in range(nd)]
            p += 4 * nd
            if cls == 1:
                return ("contiguous", addr, None)
            if cls == 2:
                return ("chunked", addr, tuple(dims[:-1]), dims[-1])
            n = int.from_bytes(b[p:p + 4], "little")
            return ("compact", bytes(b[p + 4:p + 4 + n]))
        raise H5FormatError("%s: data layout message version %d" % (self.name, ver))

    @staticmethod
    def _parse_filters(b):
        ver, n = b[0], b[1]
        p = 8 if ver == 1 else 2
        out = []
        for _ in range(n):
            fid = int.from_bytes(b[p:p + 2], "little")
            p += 2
            nlen = 0
            if ver == 1 or fid >= 256:
                nlen = int.from_bytes(b[p:p + 2], "little")
                p += 2
            p += 2   # flags
            ncd = int.from_bytes(b[p:p + 2], "little")
            p += 2
            if ver == 1:
                nlen = (nlen + 7) // 8 * 8
            p += nlen
            cd = [int.from_bytes(b[p + 4 * i:p + 4 * i + 4], "little") for i in range(ncd)]
            p += 4 * ncd
            if ver == 1 and ncd % 2:
                p += 4
            out.append((fid, cd))
        return out

    # -- data --
    def __len__(self):
        return self.shape[0]

    @property
    def ndim(self):
        return len(self.shape)

    @property
    def size(self):
        return int(np.prod(self.shape, dtype=np.int64))

    def __array__(self, dtype=None, copy=None):
        a = self.read()
        return a if dtype is None else a.astype(dtype)

    def __getitem__(self, key):
        if key is Ellipsis or (isinstance(key, tuple) and len(key) == 0):
            return self.read()
        if isinstance(key, slice) and self.shape:
            lo, hi, step = key.indices(self.shape[0])
            if step == 1:
                return self.read(lo, max(lo, hi))
        if isinstance(key, (int, np.integer)) and self.shape:
            i = int(key) + (self.shape[0] if key < 0 else 0)
            if not 0 <= i < self.shape[0]:
                raise IndexError("index %d out of range for axis 0 of size %d" % (key, self.shape[0]))
            return self.read(i, i + 1)[0]
        return self.read()[key]

    def _unfilter(self, raw, mask, nbytes):
        for i in range(len(self._filters) - 1, -1, -1):
            if mask is not None and (mask >> i) & 1:
                continue
            fid, _ = self._filters[i]
            if fid == 1:
                raw = zlib.decompress(raw)
            elif fid == 2:
                es = self.dtype.itemsize
                n = len(raw) // es
                a = np.frombuffer(raw, dtype=np.uint8, count=n * es).reshape(es, n).T
                raw = np.ascontiguousarray(a).tobytes() + bytes(raw[n * es:])
            elif fid == 3:
                raw = raw[:-4]
            else:
                raise H5FormatError("%s: filter id %d is not supported (deflate, shuffle, fletcher32 only)" % (self.name, fid))
        if len(raw) < nbytes:
            raise H5FormatError("%s: chunk holds %d bytes, expected %d" % (self.name, len(raw), nbytes))
        return raw

    def read(self, lo=None, hi=None):
        """Rows [lo:hi) of axis 0 (everything by default) as a native-endian numpy array."""
        shape = self.shape
        if not shape:
            lo = hi = None
        n0 = shape[0] if shape else 1
        lo = 0 if lo is None else lo
        hi = n0 if hi is None else hi
        out_shape = ((hi - lo,) + shape[1:]) if shape else ()
        kind = self._layout[0]
        mm = self._f._mm
        base = self._f._base
        es = self.dtype.itemsize
        row = int(np.prod(shape[1:], dtype=np.int64)) * es if shape else es
        if kind in ("contiguous", "compact"):
            if kind == "compact":
                buf = self._layout[1]
                off = 0
            else:
                addr = self._layout[1]
                if addr == UNDEF:      # never written: fill value (zeros)
                    return np.zeros(out_shape, dtype=self.dtype.newbyteorder("="))
                buf, off = mm, base + addr
            a = np.frombuffer(buf, dtype=self.dtype, count=(hi - lo) * row // es, offset=off + lo * row)
            return a.reshape(out_shape).astype(self.dtype.newbyteorder("="))
        cdims = self._layout[2]
        if len(cdims) != len(shape):
            raise H5FormatError("%s: chunk rank %d vs dataset rank %d" % (self.name, len(cdims), len(shape)))
        out = np.zeros(out_shape, dtype=self.dtype.newbyteorder("="))
        cbytes = int(np.prod(cdims, dtype=np.int64)) * es
        for offs, addr, size, mask in self._chunks(lo, hi):
            raw = mm[base + addr:base + addr + (size if size is not None else cbytes)]
            if self._filters and kind != "implicit" and size is not None:
                raw = self._unfilter(raw, mask, cbytes)
            c = np.frombuffer(raw, dtype=self.dtype, count=cbytes // es).reshape(cdims)
            src, dst = [], []
            for ax, (o, cd, sd) in enumerate(zip(offs, cdims, shape)):
                a0, a1 = o, min(o + cd, sd)
                if ax == 0:
                    a0, a1 = max(a0, lo), min(a1, hi)
                    dst.append(slice(a0 - lo, a1 - lo))
                else:
                    dst.append(slice(a0, a1))
                src.append(slice(a0 - o, a1 - o))
            if all(s.stop > s.start for s in src):
                out[tuple(dst)] = c[tuple(src)]
        return out

    def _chunks(self, lo, hi):
        kind = self._layout[0]
        cdims = self._layout[2]
        if kind == "single":
            yield (0,) * len(cdims), self._layout[1], self._layout[4], self._layout[5]
            return
        if kind == "implicit":
            counts = [-(-s // c) for s, c in zip(self.shape, cdims)]
            cbytes = int(np.prod(cdims, dtype=np.int64)) * self.dtype.itemsize
            for i, idx in enumerate(np.ndindex(*counts)):
                offs = tuple(k * c for k, c in zip(idx, cdims))
                if offs[0] < hi and offs[0] + cdims[0] > lo:
                    yield offs, self._layout[1] + i * cbytes, cbytes, 0
            return
        root = self._layout[1]
        if root == UNDEF:
            return
        if kind == "farray":
            yield from self._fixed_array(root, lo, hi)
            return
        if kind == "earray":
            yield from self._ext_array(root, lo, hi)
            return
        if kind == "btree2":
            yield from self._btree2(root, lo, hi)
            return
        yield from self._walk(root, lo, hi)

    # -- extensible array (HDF5 File Format Specification 3.0, VII.D; libhdf5 H5EA*.c) ------------------------------
    # Element i lives: in the index block for i < idx_blk_elmts; otherwise in a data block of super block u, which holds
    # 2^(u/2) data blocks of dblk_min * 2^((u+1)/2) elements.  The index block addresses the data blocks of the first
    # 2*log2(sup_blk_min_data_ptrs) super blocks directly, the other super blocks through EASB blocks.  A data block with
    # more than 2^page_bits elements keeps them in pages (each with its own checksum) behind its prefix.
    def _ext_array(self, addr, lo, hi):
        mm, base = self._f._mm, self._f._base
        u64 = lambda q, n=8: int.from_bytes(mm[q:q + n], "little")
        h = base + addr
        if mm[h:h + 4] != b"EAHD" or mm[h + 4] != 0:
            raise H5FormatError("%s: bad extensible-array header" % self.name)
        filtered, esz, max_bits, idx_elmts = mm[h + 5] == 1, mm[h + 6], mm[h + 7], mm[h + 8]
        dblk_min, sup_min_ptrs, page_bits = mm[h + 9], mm[h + 10], mm[h + 11]
        max_idx_set = u64(h + 12 + 8 * 4)                       # statistics: 6 lengths, the 5th = highest index set + 1
        iblk = u64(h + 12 + 8 * 6)
        if iblk == UNDEF:
            return
        ib = base + iblk
        if mm[ib:ib + 4] != b"EAIB":
            raise H5FormatError("%s: bad extensible-array index block" % self.name)
        off_size = (max_bits + 7) // 8
        log2 = lambda x: x.bit_length() - 1
        nsblks = 1 + (max_bits - log2(dblk_min))
        ib_nsblks = 2 * log2(sup_min_ptrs)
        ndblk_addrs = 2 * (sup_min_ptrs - 1)
        per_page = 1 << page_bits
        cdims = self._layout[2]
        cbytes = int(np.prod(cdims, dtype=np.int64)) * self.dtype.itemsize
        # chunk coordinates of a linear index: the unlimited dimension is the slowest one ("swizzled" when it is not dim 0)
        rank = len(self.shape)
        unlim = [i for i, m in enumerate(self.maxshape) if m == UNDEF]
        if len(unlim) != 1:
            raise H5FormatError("%s: extensible-array index without exactly one unlimited dimension" % self.name)
        ud = unlim[0]
        order = [ud] + [i for i in range(rank) if i != ud]
        maxc = [-(-self.maxshape[i] // cdims[i]) if self.maxshape[i] != UNDEF else None for i in range(rank)]
        down = {}
        acc = 1
        for i in reversed(order):
            down[i] = acc
            acc *= (maxc[i] if maxc[i] is not None else 1)

        def coords(i):
            out = [0] * rank
            for d in order:
                out[d], i = divmod(i, down[d])
            return tuple(k * c for k, c in zip(out, cdims))

        def emit(i, e):
            caddr = u64(e)
            if caddr == UNDEF or i >= max_idx_set:
                return None
            offs = coords(i)
            if any(o >= s for o, s in zip(offs, self.shape)) or not (offs[0] < hi and offs[0] + cdims[0] > lo):
                return None
            if filtered:
                return offs, caddr, u64(e + 8, esz - 12), u64(e + esz - 4, 4)
            return offs, caddr, cbytes, 0

        def data_block(daddr, first, nelmts):
            d = base + daddr
            if mm[d:d + 4] != b"EADB":
                raise H5FormatError("%s: bad extensible-array data block" % self.name)
            q = d + 6 + 8 + off_size                         # signature, version, client, header address, block offset
            if nelmts > per_page:                            # paged: prefix checksum, then pages of per_page elements + checksum
                q += 4
                for i in range(nelmts):
                    yield first + i, q + i * esz + (i // per_page) * 4
            else:
                for i in range(nelmts):
                    yield first + i, q + i * esz

        q = ib + 6 + 8
        for i in range(idx_elmts):
            r = emit(i, q + i * esz)
            if r:
                yield r
        q += idx_elmts * esz
        dblk_addrs = [u64(q + 8 * i) for i in range(ndblk_addrs)]
        q += 8 * ndblk_addrs
        sblk_addrs = [u64(q + 8 * i) for i in range(max(nsblks - ib_nsblks, 0))]
        start, k = idx_elmts, 0
        for u in range(nsblks):
            ndblks, dn = 1 << (u // 2), dblk_min << ((u + 1) // 2)
            if start >= max_idx_set:
                break
            if u < ib_nsblks:
                addrs = dblk_addrs[k:k + ndblks]
                k += ndblks
            else:
                sa = sblk_addrs[u - ib_nsblks]
                addrs = [UNDEF] * ndblks
                if sa != UNDEF:
                    sb = base + sa
                    if mm[sb:sb + 4] != b"EASB":
                        raise H5FormatError("%s: bad extensible-array super block" % self.name)
                    p2 = sb + 6 + 8 + off_size
                    if dn > per_page:                        # page-initialisation bitmaps of its (paged) data blocks
                        p2 += ndblks * ((dn // per_page + 7) // 8)
                    addrs = [u64(p2 + 8 * j) for j in range(ndblks)]
            for j, da in enumerate(addrs):
                if da != UNDEF:
                    for i, e in data_block(da, start + j * dn, dn):
                        r = emit(i, e)
                        if r:
                            yield r
            start += ndblks * dn

    # -- v2 B-tree chunk index (specification III.A.2, record types 10 / 11; libhdf5 H5B2*.c, H5Dbtree2.c) ------------
    def _btree2(self, addr, lo, hi):
        mm, base = self._f._mm, self._f._base
        u64 = lambda q, n=8: int.from_bytes(mm[q:q + n], "little")
        h = base + addr
        if mm[h:h + 4] != b"BTHD" or mm[h + 4] != 0:
            raise H5FormatError("%s: bad v2 B-tree header" % self.name)
        rtype, node_size, rec_size, depth = mm[h + 5], u64(h + 6, 4), u64(h + 10, 2), u64(h + 12, 2)
        if rtype not in (10, 11):
            raise H5FormatError("%s: v2 B-tree of record type %d is not a chunk index" % (self.name, rtype))
        root, root_nrec = u64(h + 16), u64(h + 24, 2)
        if root == UNDEF or root_nrec == 0:
            return
        rank = len(self.shape)
        cdims = self._layout[2]
        cbytes = int(np.prod(cdims, dtype=np.int64)) * self.dtype.itemsize
        enc = lambda limit: (max(int(limit), 1).bit_length() - 1) // 8 + 1          # H5VM_limit_enc_size
        max_nrec = [(node_size - 10) // rec_size]                                    # per depth: leaf first
        cum_max, cum_size = [max_nrec[0]], [0]
        nrec_size = enc(max_nrec[0])
        for d in range(1, depth + 1):
            ptr = 8 + nrec_size + cum_size[d - 1]
            max_nrec.append((node_size - (10 + ptr)) // (rec_size + ptr))
            cum_max.append((max_nrec[d] + 1) * cum_max[d - 1] + max_nrec[d])
            cum_size.append(enc(cum_max[d]))

        def record(q):
            caddr = u64(q)
            if rtype == 11:
                size_len = rec_size - 8 - 4 - 8 * rank
                size, mask, q2 = u64(q + 8, size_len), u64(q + 8 + size_len, 4), q + 12 + size_len
            else:
                size, mask, q2 = cbytes, 0, q + 8
            offs = tuple(u64(q2 + 8 * i) * cdims[i] for i in range(rank))            # scaled (chunk) coordinates
            if caddr != UNDEF and offs[0] < hi and offs[0] + cdims[0] > lo:
                return offs, caddr, size, mask
            return None

        def node(naddr, nrec, d):
            n = base + naddr
            sig = b"BTIN" if d > 0 else b"BTLF"
            if mm[n:n + 4] != sig:
                raise H5FormatError("%s: bad v2 B-tree node" % self.name)
            q = n + 6
            for i in range(nrec):
                r = record(q + i * rec_size)
                if r:
                    yield r
            if d > 0:
                q += nrec * rec_size
                ptr = 8 + nrec_size + cum_size[d - 1]
                for i in range(nrec + 1):
                    c = q + i * ptr
                    yield from node(u64(c), u64(c + 8, nrec_size), d - 1)

        yield from node(root, root_nrec, depth)

    def _fixed_array(self, addr, lo, hi):
        mm, base = self._f._mm, self._f._base
        h = base + addr
        if mm[h:h + 4] != b"FAHD" or mm[h + 4] != 0:
            raise H5FormatError("%s: bad fixed-array header" % self.name)
        filtered, esz, page_bits = mm[h + 5] == 1, mm[h + 6], mm[h + 7]
        nent = int.from_bytes(mm[h + 8:h + 16], "little")
        db = int.from_bytes(mm[h + 16:h + 24], "little")
        if db == UNDEF:
            return
        d = base + db
        if mm[d:d + 4] != b"FADB":
            raise H5FormatError("%s: bad fixed-array data block" % self.name)
        cdims = self._layout[2]
        counts = [-(-s // c) for s, c in zip(self.shape, cdims)]
        cbytes = int(np.prod(cdims, dtype=np.int64)) * self.dtype.itemsize
        per_page = 1 << page_bits
        q = d + 14
        paged = nent > per_page
        if paged:
            npages = -(-nent // per_page)
            q += (npages + 7) // 8 + 4      # page-init bitmap, then the data block's own checksum; pages follow
        for i, idx in enumerate(np.ndindex(*counts)):
            if i >= nent:
                break
            e = q + i * esz + ((i // per_page) * 4 if paged else 0)      # every page ends with a 4-byte checksum
            caddr = int.from_bytes(mm[e:e + 8], "little")
            if caddr == UNDEF:
                continue
            offs = tuple(k * c for k, c in zip(idx, cdims))
            if not (offs[0] < hi and offs[0] + cdims[0] > lo):
                continue
            if filtered:
                size = int.from_bytes(mm[e + 8:e + esz - 4], "little")
                mask = int.from_bytes(mm[e + esz - 4:e + esz], "little")
                yield offs, caddr, size, mask
            else:
                yield offs, caddr, cbytes, 0

    def _walk(self, addr, lo, hi):
        mm, base = self._f._mm, self._f._base
        p = base + addr
        if mm[p:p + 4] != b"TREE" or mm[p + 4] != 1:
            raise H5FormatError("%s: bad chunk B-tree node at %d" % (self.name, addr))
        level = mm[p + 5]
        used = int.from_bytes(mm[p + 6:p + 8], "little")
        nd = len(self.shape) + 1
        ksize = 8 + 8 * nd
        q = p + 24
        for _ in range(used):
            size = int.from_bytes(mm[q:q + 4], "little")
            mask = int.from_bytes(mm[q + 4:q + 8], "little")
            offs = tuple(int.from_bytes(mm[q + 8 + 8 * i:q + 16 + 8 * i], "little") for i in range(nd - 1))
            child = int.from_bytes(mm[q + ksize:q + ksize + 8], "little")
            nxt = q + ksize + 8
            if level == 0:
                if offs[0] < hi and offs[0] + self._layout[2][0] > lo:
                    yield offs, child, size, mask
            elif offs[0] < hi:     # the key is the first chunk of the subtree
                yield from self._walk(child, lo, hi)
            q = nxt


class Group:
    def __init__(self, f, name, links):
        self._f, self.name, self._links = f, name, links

    def keys(self):
        return list(self._links)

    def __iter__(self):
        return iter(self._links)

    def __len__(self):
        return len(self._links)

    def __contains__(self, k):
        try:
            self[k]
            return True
        except (KeyError, H5FormatError):
            return False

    def __getitem__(self, path):
        node = self
        for part in [p for p in path.split("/") if p]:
            if not isinstance(node, Group) or part not in node._links:
                raise KeyError(path)
            node = node._f._open(node._links[part], (node.name.rstrip("/") + "/" + part))
        return node


class H5File(Group):
    """``with H5File(path) as f: f['src'][l:r]`` -- read-only."""

    def __init__(self, path, mode="r"):
        if mode != "r":
            raise ValueError("H5File is read-only; use write_h5() to create files")
        self._fh = open(path, "rb")
        try:
            self._mm = mmap.mmap(self._fh.fileno(), 0, access=mmap.ACCESS_READ)
        except ValueError:
            self._fh.close()
            raise H5FormatError("%s: empty file" % path)
        self._cache = {}
        root = self._superblock(path)
        g = self._open(root, "/")
        if not isinstance(g, Group):
            raise H5FormatError("%s: root object is not a group" % path)
        Group.__init__(self, self, "/", g._links)

    def close(self):
        if self._mm is not None:
            self._mm.close()
            self._fh.close()
            self._mm = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _superblock(self, path):
        mm = self._mm
        off = 0
        while True:     # the superblock may sit at 0, 512, 1024, ... (user block)
            if mm[off:off + 8] == SIGNATURE:
                break
            off = 512 if off == 0 else off * 2
            if off + 8 > len(mm):
                raise H5FormatError("%s: not an HDF5 file" % path)
        ver = mm[off + 8]
        if ver in (0, 1):
            if mm[off + 13] != 8 or mm[off + 14] != 8:
                raise H5FormatError("only 8-byte offsets / lengths are supported")
            p = off + 24 + (4 if ver == 1 else 0)
            self._base = int.from_bytes(mm[p:p + 8], "little")
            entry = p + 32
            return int.from_bytes(mm[entry + 8:entry + 16], "little")
        if ver in (2, 3):
            if mm[off + 9] != 8 or mm[off + 10] != 8:
                raise H5FormatError("only 8-byte offsets / lengths are supported")
            self._base = int.from_bytes(mm[off + 12:off + 20], "little")
            return int.from_bytes(mm[off + 36:off + 44], "little")
        raise H5FormatError("superblock version %d" % ver)

    # -- object headers --
    def _messages(self, addr):
        mm, p = self._mm, self._base + addr
        msgs = []
        if mm[p:p + 4] == b"OHDR":
            if mm[p + 4] != 2:
                raise H5FormatError("object header version %d" % mm[p + 4])
            flags = mm[p + 5]
            q = p + 6
            if flags & 0x20:
                q += 16
            if flags & 0x10:
                q += 4
            nsz = 1 << (flags & 3)
            size0 = int.from_bytes(mm[q:q + nsz], "little")
            q += nsz
            blocks = [(q, q + size0)]
            corder = bool(flags & 0x04)
            while blocks:
                q, end = blocks.pop(0)
                while q + 4 <= end:
                    typ = mm[q]
                    size = int.from_bytes(mm[q + 1:q + 3], "little")
                    q += 4 + (2 if corder else 0)
                    data = mm[q:q + size]
                    if typ == 0x10:
                        caddr = int.from_bytes(data[0:8], "little")
                        clen = int.from_bytes(data[8:16], "little")
                        c = self._base + caddr
                        if mm[c:c + 4] != b"OCHK":
                            raise H5FormatError("bad object header continuation at %d" % caddr)
                        blocks.append((c + 4, c + clen - 4))
                    elif typ != 0:
                        msgs.append((typ, data))
                    q += size
            return msgs
        ver = mm[p]
        if ver != 1:
            raise H5FormatError("object header version %d at %d" % (ver, addr))
        nmsg = int.from_bytes(mm[p + 2:p + 4], "little")
        size0 = int.from_bytes(mm[p + 8:p + 12], "little")
        blocks = [(p + 16, p + 16 + size0)]
        while blocks and nmsg > 0:
            q, end = blocks.pop(0)
            while q + 8 <= end and nmsg > 0:
                typ = int.from_bytes(mm[q:q + 2], "little")
                size = int.from_bytes(mm[q + 2:q + 4], "little")
                data = mm[q + 8:q + 8 + size]
                nmsg -= 1
                if typ == 0x10:
                    caddr = int.from_bytes(data[0:8], "little")
                    clen = int.from_bytes(data[8:16], "little")
                    blocks.append((self._base + caddr, self._base + caddr + clen))
                elif typ != 0:
                    msgs.append((typ, data))
                q += 8 + size
        return msgs

    def _open(self, addr, name):
        if addr in self._cache:
            return self._cache[addr]
        msgs = self._messages(addr)
        types = {t for t, _ in msgs}
        if 0x0008 in types:
            obj = Dataset(self, name, msgs)
        else:
            links = {}
            for typ, data in msgs:
                if typ == 0x0011:
                    bt = int.from_bytes(data[0:8], "little")
                    heap = int.from_bytes(data[8:16], "little")
                    self._symbols(bt, heap, links)
                elif typ == 0x0006:
                    k, v = self._link(data)
                    if k is not None:
                        links[k] = v
                elif typ == 0x0002:
                    ver, fl = data[0], data[1]
                    q = 2 + (8 if fl & 1 else 0)
                    fheap = int.from_bytes(data[q:q + 8], "little")
                    if fheap != UNDEF:
                        self._dense_links(fheap, links, name)
            obj = Group(self, name, links)
        self._cache[addr] = obj
        return obj

    @staticmethod
    def _link(b, want_len=False):
        if b[0] != 1:
            raise H5FormatError("link message version %d" % b[0])
        fl = b[1]
        p = 2
        ltype = 0
        if fl & 0x08:
            ltype = b[p]
            p += 1
        if fl & 0x04:
            p += 8
        if fl & 0x10:
            p += 1
        nsz = 1 << (fl & 3)
        n = int.from_bytes(b[p:p + nsz], "little")
        p += nsz
        name = bytes(b[p:p + n]).decode("utf-8")
        p += n
        if ltype != 0:             # soft / external links: skipped
            if ltype == 1:
                p += 2 + int.from_bytes(b[p:p + 2], "little")
            elif want_len:
                raise H5FormatError("user-defined link in dense storage")
            return (None, None, p) if want_len else (None, None)
        addr = int.from_bytes(b[p:p + 8], "little")
        return (name, addr, p + 8) if want_len else (name, addr)

    def _dense_links(self, fheap, links, name):
        """Dense link storage: the link messages live as managed objects of a fractal heap.  They are read by scanning
        the heap's direct blocks front to back, which is how an append-only file lays them out; a heap whose object
        count disagrees with the scan (objects were deleted) is refused."""
        mm, h = self._mm, self._base + fheap
        if mm[h:h + 4] != b"FRHP" or mm[h + 4] != 0:
            raise H5FormatError("%s: bad fractal heap header" % name)
        if int.from_bytes(mm[h + 7:h + 9], "little") != 0:
            raise H5FormatError("%s: filtered fractal heap" % name)
        flags = mm[h + 9]
        nobj = int.from_bytes(mm[h + 70:h + 78], "little")
        if int.from_bytes(mm[h + 86:h + 94], "little") or int.from_bytes(mm[h + 102:h + 110], "little"):
            raise H5FormatError("%s: huge / tiny fractal-heap objects" % name)
        width = int.from_bytes(mm[h + 110:h + 112], "little")
        start = int.from_bytes(mm[h + 112:h + 120], "little")
        max_direct = int.from_bytes(mm[h + 120:h + 128], "little")
        max_bits = int.from_bytes(mm[h + 128:h + 130], "little")
        root = int.from_bytes(mm[h + 132:h + 140], "little")
        rows = int.from_bytes(mm[h + 140:h + 142], "little")
        off_bytes = (max_bits + 7) // 8
        blocks = []
        if rows == 0:
            blocks.append((root, start))
        else:
            r = self._base + root
            if mm[r:r + 4] != b"FHIB":
                raise H5FormatError("%s: bad fractal heap indirect block" % name)
            q = r + 13 + off_bytes
            for row in range(rows):
                size = start if row < 2 else start << (row - 1)
                if size > max_direct:
                    raise H5FormatError("%s: nested fractal-heap indirect blocks" % name)
                for _ in range(width):
                    addr = int.from_bytes(mm[q:q + 8], "little")
                    q += 8
                    if addr != UNDEF:
                        blocks.append((addr, size))
        found = 0
        for addr, size in blocks:
            d = self._base + addr
            if mm[d:d + 4] != b"FHDB":
                raise H5FormatError("%s: bad fractal heap direct block" % name)
            q = d + 13 + off_bytes + (4 if flags & 0x02 else 0)
            end = d + size
            while q < end and mm[q] == 1:
                k, v, n = self._link(mm[q:end], want_len=True)
                if k is not None:
                    links[k] = v
                found += 1
                q += n
        if found != nobj:
            raise H5FormatError("%s: fractal heap holds %d objects, scan found %d (deleted links?)" % (name, nobj, found))

    def _symbols(self, bt, heap, links):
        mm, base = self._mm, self._base
        h = base + heap
        if mm[h:h + 4] != b"HEAP":
            raise H5FormatError("bad local heap at %d" % heap)
        hdata = base + int.from_bytes(mm[h + 24:h + 32], "little")

        def name_at(off):
            e = mm.find(b"\0", hdata + off)
            return bytes(mm[hdata + off:e]).decode("utf-8")

        def node(addr):
            p = base + addr
            if mm[p:p + 4] == b"SNOD":
                n = int.from_bytes(mm[p + 6:p + 8], "little")
                for i in range(n):
                    e = p + 8 + 40 * i
                    links[name_at(int.from_bytes(mm[e:e + 8], "little"))] = int.from_bytes(mm[e + 8:e + 16], "little")
                return
            if mm[p:p + 4] != b"TREE" or mm[p + 4] != 0:
                raise H5FormatError("bad group B-tree node at %d" % addr)
            used = int.from_bytes(mm[p + 6:p + 8], "little")
            for i in range(used):
                node(int.from_bytes(mm[p + 24 + 8 + 16 * i:p + 24 + 16 + 16 * i], "little"))

        node(bt)


# ---------------------------------------------------------------------------------------------------------------------
# writer
# ---------------------------------------------------------------------------------------------------------------------
def _pad8(b):
    return b + b"\0" * (-len(b) % 8)


def _msg(typ, data, flags=0):
    data = _pad8(data)
    return struct.pack("<HHB3x", typ, len(data), flags) + data


def _datatype_msg(dt):
    dt = np.dtype(dt)
    if dt.kind in "iu" and dt.itemsize in (1, 2, 4, 8):
        bits0 = 0x08 if dt.kind == "i" else 0x00
        return struct.pack("<BBBBI", 0x10, bits0, 0, 0, dt.itemsize) + struct.pack("<HH", 0, dt.itemsize * 8)
    if dt.kind == "f" and dt.itemsize in (4, 8):
        eloc, esize, msize, bias = ((23, 8, 23, 127) if dt.itemsize == 4 else (52, 11, 52, 1023))
        return (struct.pack("<BBBBI", 0x11, 0x20, dt.itemsize * 8 - 1, 0, dt.itemsize) +
                struct.pack("<HHBBBBI", 0, dt.itemsize * 8, eloc, esize, 0, msize, bias))
    raise H5FormatError("write_h5: dtype %s is not supported (integers, float32, float64)" % dt)


def write_h5(path, datasets):
    """Create ``path`` with one contiguous root-level dataset per item of ``datasets`` ({name: array-like})."""
    items = []
    for name, a in datasets.items():
        a = np.asarray(a)
        if a.dtype.byteorder == ">" or (a.dtype.byteorder == "=" and not np.little_endian):
            a = a.astype(a.dtype.newbyteorder("<"))
        nb = name.encode("utf-8")
        if not nb or b"/" in nb or b"\0" in nb:
            raise ValueError("write_h5: bad dataset name %r" % name)
        items.append((nb, np.array(a, order="C", copy=True)))      # (np.ascontiguousarray would promote 0-d to 1-d)
    items.sort(key=lambda t: t[0])     # symbol-table entries are ordered by name (strcmp)
    if len(items) > 64:
        raise ValueError("write_h5: at most 64 datasets")
    leaf_k = max(4, (len(items) + 1) // 2)       # one symbol node holds 2*leaf_k entries
    internal_k = 16

    # local heap data segment: "" at 0, then the names
    heap = bytearray(b"\0" * 8)
    name_off = []
    for nb, _ in items:
        name_off.append(len(heap))
        heap += _pad8(nb + b"\0")
    free_off = len(heap)
    heap += struct.pack("<QQ", 1, 32) + b"\0" * 16    # one free block (next = H5HL_FREE_NULL, size 32)

    sb_size = 96
    root_oh = sb_size
    root_msgs_len = 8 + 16
    btree = root_oh + 16 + root_msgs_len
    btree_size = 24 + (2 * internal_k + 1) * 8 + 2 * internal_k * 8
    heap_hdr = btree + btree_size
    heap_data = heap_hdr + 32
    snod = heap_data + len(heap)
    snod_size = 8 + 2 * leaf_k * 40
    pos = snod + snod_size

    # dataset object headers, then raw data (8-byte aligned)
    headers, oh_addr = [], []
    for nb, a in items:
        space = struct.pack("<BBBB4x", 1, a.ndim, 0, 0) + b"".join(struct.pack("<Q", d) for d in a.shape)
        msgs = (_msg(0x0001, space) + _msg(0x0003, _datatype_msg(a.dtype), flags=1) +
                _msg(0x0005, struct.pack("<BBBB", 2, 2, 2, 0)))
        layout_at = len(msgs) + 8      # offset of the layout message's data inside the message block
        msgs += _msg(0x0008, struct.pack("<BBQQ", 3, 1, 0, a.nbytes))
        headers.append([msgs, layout_at])
        oh_addr.append(pos)
        pos += 16 + len(msgs)
    data_addr = []
    for _, a in items:
        pos = (pos + 7) // 8 * 8
        data_addr.append(pos if a.nbytes else UNDEF)
        pos += a.nbytes
    eof = pos

    out = bytearray()
    # superblock v0
    out += SIGNATURE + struct.pack("<BBBBBBBB", 0, 0, 0, 0, 0, 8, 8, 0) + struct.pack("<HHI", leaf_k, internal_k, 0)
    out += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
    out += struct.pack("<QQI4xQQ", 0, root_oh, 1, btree, heap_hdr)      # root symbol-table entry, cached B-tree / heap
    assert len(out) == sb_size
    # root group object header (v1): one symbol-table message
    out += struct.pack("<BBHII4x", 1, 0, 1, 1, root_msgs_len) + _msg(0x0011, struct.pack("<QQ", btree, heap_hdr))
    assert len(out) == btree
    # group B-tree: one leaf child (the symbol node); key[0] = "" and key[1] = the largest name
    node = b"TREE" + struct.pack("<BBHQQ", 0, 0, 1 if items else 0, UNDEF, UNDEF)
    node += struct.pack("<QQQ", 0, snod, name_off[-1] if items else 0)
    out += node + b"\0" * (btree_size - len(node))
    # local heap header + data segment
    out += b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap), free_off, heap_data) + heap
    assert len(out) == snod
    # symbol node
    s = b"SNOD" + struct.pack("<BBH", 1, 0, len(items))
    for off, addr in zip(name_off, oh_addr):
        s += struct.pack("<QQI4x16x", off, addr, 0)
    out += s + b"\0" * (snod_size - len(s))
    # dataset headers
    for (msgs, layout_at), addr, daddr in zip(headers, oh_addr, data_addr):
        assert len(out) == addr
        m = bytearray(msgs)
        m[layout_at + 2:layout_at + 10] = struct.pack("<Q", daddr)
        out += struct.pack("<BBHII4x", 1, 0, 4, 1, len(m)) + m
    # raw data
    with open(path, "wb") as fh:
        fh.write(out)
        at = len(out)
        for (_, a), daddr in zip(items, data_addr):
            if not a.nbytes:
                continue
            fh.write(b"\0" * (daddr - at))
            fh.write(a.tobytes())
            at = daddr + a.nbytes
    return path
