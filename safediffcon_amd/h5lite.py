"""A minimal read-only HDF5 reader -- just enough for the Keras weight files of the KSTAR surrogate
(``tokamak/weights/{lstm/v220505,nn,bpw}/best_model*``, which ``tokamak/common/model_structure.py:69-152`` opens through
TensorFlow/h5py; neither is installed here, and the rollout does not need them).

Supported, because that is what h5py writes for such files: superblock version 0/1, version-1 object headers with
continuation blocks, old-style groups (symbol-table message -> v1 B-tree -> SNOD nodes -> local heap), contiguous and compact
little-endian integer / IEEE-float datasets, and attributes holding fixed-length strings, variable-length strings (global
heap) and numeric arrays.  Anything else (chunked or filtered data, new-style groups, v2 object headers, big-endian data)
raises ``H5Error`` naming what was met -- this is not a general HDF5 library.

    f = h5lite.File(path)
    f.attrs["model_config"]                    -> bytes
    g = f["model_weights"];  g.keys();  g.attrs["layer_names"]
    f["model_weights/dense_1/dense_1/kernel:0"][...]  -> numpy array
"""
import struct

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF


class H5Error(ValueError):
    pass


def _pad8(n):
    return (n + 7) & ~7


def _guard(fn):
    """a truncated or foreign file shows up as an out-of-range read: report it as H5Error, not as struct.error / IndexError"""
    def wrapped(self, *a, **k):
        try:
            return fn(self, *a, **k)
        except H5Error:
            raise
        except (struct.error, IndexError, OverflowError, UnicodeDecodeError, ValueError) as e:
            f = getattr(self, "f", self)
            raise H5Error(f"{getattr(f, 'path', '?')}: malformed or truncated HDF5 structure ({type(e).__name__}: {e})") from e
    wrapped.__name__, wrapped.__doc__ = fn.__name__, fn.__doc__
    return wrapped


class _Type:
    """a parsed datatype message"""

    def __init__(self, buf, off):
        cv, b0, b1, b2, size = struct.unpack_from("<BBBBI", buf, off)
        self.cls, self.version, self.size = cv & 0x0F, cv >> 4, size
        self.bits = (b0, b1, b2)
        self.base = None
        if self.cls == 0:                                    # fixed point
            if b0 & 1:
                raise H5Error("big-endian integers are not supported")
            self.dtype = np.dtype(("<i" if b0 & 8 else "<u") + str(size))
        elif self.cls == 1:                                  # IEEE float
            if b0 & 1:
                raise H5Error("big-endian floats are not supported")
            if size not in (2, 4, 8):
                raise H5Error(f"float of {size} bytes")
            self.dtype = np.dtype("<f" + str(size))
        elif self.cls == 3:                                  # fixed-length string
            self.dtype = np.dtype("S" + str(size))
        elif self.cls == 9:                                  # variable length (sequence or string) of a base type
            self.is_vlen_string = (b0 & 0x0F) == 1
            self.base = _Type(buf, off + 8)
            self.dtype = None
        else:
            raise H5Error(f"datatype class {self.cls} is not supported")


def _dataspace(buf, off, L):
    version, rank, flags = struct.unpack_from("<BBB", buf, off)
    if version == 1:
        p = off + 8
    elif version == 2:
        p = off + 4
    else:
        raise H5Error(f"dataspace message version {version}")
    fmt = "<" + ("Q" if L == 8 else "I") * rank
    return tuple(struct.unpack_from(fmt, buf, p)) if rank else ()


class _Node:
    """an object header: messages parsed on demand"""

    def __init__(self, f, addr):
        self.f, self.addr = f, addr
        self.msgs = f._messages(addr)

    def _find(self, mtype):
        return [(off, size) for (t, off, size) in self.msgs if t == mtype]

    @property
    def attrs(self):
        out = {}
        for off, size in self._find(0x000C):
            name, value = self.f._attribute(off)
            out[name] = value
        return out


class Dataset(_Node):
    def __init__(self, f, addr):
        super().__init__(f, addr)
        buf, L = f.buf, f.L
        (toff, _), = self._find(0x0003)
        (soff, _), = self._find(0x0001)
        self.type = _Type(buf, toff)
        self.shape = _dataspace(buf, soff, L)
        if self._find(0x000B):
            raise H5Error("filtered (compressed) datasets are not supported")
        (loff, _), = self._find(0x0008)
        version = buf[loff]
        if version == 3:
            cls = buf[loff + 1]
            if cls == 1:
                self.data_addr, self.data_size = f._unpack_OL(loff + 2)
            elif cls == 0:
                (n,) = struct.unpack_from("<H", buf, loff + 2)
                self.data_addr, self.data_size = loff + 4, n
            else:
                raise H5Error("chunked datasets are not supported")
        elif version in (1, 2):
            rank, cls = buf[loff + 1], buf[loff + 2]
            if cls != 1:
                raise H5Error("only contiguous data in layout message versions 1 / 2")
            (self.data_addr,) = f._unpack_O(loff + 8)
            dims = struct.unpack_from("<" + "I" * rank, buf, loff + 8 + f.O)
            self.data_size = int(np.prod(dims))
        else:
            raise H5Error(f"data layout message version {version}")
        self.dtype = self.type.dtype
        if self.dtype is None:
            raise H5Error("variable-length datasets are not supported")

    @_guard
    def __getitem__(self, key):
        n = int(np.prod(self.shape)) if self.shape else 1
        if self.data_addr == UNDEF:                          # never written: the fill value, which h5py leaves at zero
            arr = np.zeros(self.shape, dtype=self.dtype)
        else:
            arr = np.frombuffer(self.f.buf, dtype=self.dtype, count=n, offset=self.data_addr).reshape(self.shape).copy()
        return arr[key]


class Group(_Node):
    def __init__(self, f, addr, btree=None, heap=None):
        super().__init__(f, addr)
        if btree is None:
            st = self._find(0x0011)
            if not st:
                raise H5Error("new-style (link message) groups are not supported")
            btree, heap = f._unpack_OO(st[0][0])
        self._links = f._group_links(btree, heap)

    def keys(self):
        return list(self._links)

    def __contains__(self, name):
        return name in self._links

    def __getitem__(self, path):
        node = self
        for part in path.strip("/").split("/"):
            if not isinstance(node, Group):
                raise KeyError(path)
            if part not in node._links:
                raise KeyError(f"{path!r}: no member {part!r} (have {sorted(node._links)})")
            node = node.f._open(node._links[part])
        return node

    def visit_datasets(self, prefix=""):
        """-> {path: Dataset} of everything below this group"""
        out = {}
        for name in self._links:
            node = self[name]
            p = f"{prefix}{name}"
            if isinstance(node, Group):
                out.update(node.visit_datasets(p + "/"))
            else:
                out[p] = node
        return out


class File(Group):
    def __init__(self, path):
        with open(path, "rb") as fh:
            self.buf = fh.read()
        self.path = str(path)
        self._parse_superblock()

    @_guard
    def _parse_superblock(self):
        path, buf = self.path, self.buf
        if buf[:8] != b"\x89HDF\r\n\x1a\n":
            raise H5Error(f"{path}: not an HDF5 file")
        version = buf[8]
        if version not in (0, 1):
            raise H5Error(f"{path}: superblock version {version} is not supported")
        self.O, self.L = buf[13], buf[14]
        if self.O not in (4, 8) or self.L not in (4, 8):
            raise H5Error("offset / length sizes must be 4 or 8")
        p = 24 + (4 if version == 1 else 0)
        (self.base,) = self._unpack_O(p)
        if self.base != 0:
            raise H5Error("a non-zero base address is not supported")
        p += 4 * self.O                                       # base, free-space, end-of-file, driver-info addresses
        # root group symbol table entry: link name offset, object header address, cache type, reserved, scratch pad
        (hdr,) = self._unpack_O(p + self.O)
        (cache,) = struct.unpack_from("<I", buf, p + 2 * self.O)
        self._cache = {}
        if cache == 1:
            bt, hp = self._unpack_OO(p + 2 * self.O + 8)
            Group.__init__(self, self, hdr, bt, hp)
        else:
            Group.__init__(self, self, hdr)

    # ---- primitives
    def _unpack_O(self, off):
        return struct.unpack_from("<Q" if self.O == 8 else "<I", self.buf, off)

    def _unpack_OO(self, off):
        return struct.unpack_from("<QQ" if self.O == 8 else "<II", self.buf, off)

    def _unpack_OL(self, off):
        a, = self._unpack_O(off)
        b, = struct.unpack_from("<Q" if self.L == 8 else "<I", self.buf, off + self.O)
        return a, b

    def _messages(self, addr):
        buf = self.buf
        version = buf[addr]
        if buf[addr:addr + 4] == b"OHDR":
            raise H5Error("version-2 object headers are not supported")
        if version != 1:
            raise H5Error(f"object header version {version} at {addr:#x}")
        nmsg, = struct.unpack_from("<H", buf, addr + 2)
        hsize, = struct.unpack_from("<I", buf, addr + 8)
        blocks = [(addr + 16, hsize)]
        msgs = []
        while blocks and len(msgs) < nmsg:
            p, left = blocks.pop(0)
            end = p + left
            while p + 8 <= end and len(msgs) < nmsg:
                mtype, msize, _flags = struct.unpack_from("<HHB", buf, p)
                body = p + 8
                if mtype == 0x0010:                          # continuation
                    caddr, clen = self._unpack_OL(body)
                    blocks.append((caddr, clen))
                msgs.append((mtype, body, msize))
                p = body + msize
        return msgs

    def _local_heap_data(self, heap):
        if self.buf[heap:heap + 4] != b"HEAP":
            raise H5Error(f"no local heap at {heap:#x}")
        (seg,) = self._unpack_O(heap + 8 + 2 * self.L)
        return seg

    def _group_links(self, btree, heap):
        seg = self._local_heap_data(heap)
        links = {}

        def name_at(off):
            a = seg + off
            return self.buf[a:self.buf.index(b"\x00", a)].decode("utf-8")

        def walk(node):
            buf = self.buf
            sig = buf[node:node + 4]
            if sig == b"TREE":
                ntype, level, used = struct.unpack_from("<BBH", buf, node + 4)
                if ntype != 0:
                    raise H5Error("a chunk B-tree where a group B-tree was expected")
                p = node + 8 + 2 * self.O                     # past the sibling addresses
                for i in range(used):
                    p += self.L                               # key i
                    (child,) = self._unpack_O(p)
                    p += self.O
                    walk(child)
            elif sig == b"SNOD":
                (nsym,) = struct.unpack_from("<H", buf, node + 6)
                p = node + 8
                esize = 2 * self.O + 24
                for i in range(nsym):
                    noff, = self._unpack_O(p)
                    hdr, = self._unpack_O(p + self.O)
                    links[name_at(noff)] = hdr
                    p += esize
            else:
                raise H5Error(f"unexpected node signature {sig!r} at {node:#x}")

        walk(btree)
        return links

    @_guard
    def _open(self, addr):
        if addr not in self._cache:
            types = {t for (t, _, _) in self._messages(addr)}
            self._cache[addr] = Dataset(self, addr) if 0x0008 in types else Group(self, addr)
        return self._cache[addr]

    def _global_heap_object(self, coll, index):
        buf = self.buf
        if buf[coll:coll + 4] != b"GCOL":
            raise H5Error(f"no global heap collection at {coll:#x}")
        size, = struct.unpack_from("<Q" if self.L == 8 else "<I", buf, coll + 8)
        p, end = coll + 8 + self.L, coll + size
        while p + 8 + self.L <= end:
            idx, = struct.unpack_from("<H", buf, p)
            osize, = struct.unpack_from("<Q" if self.L == 8 else "<I", buf, p + 8)
            if idx == index:
                return buf[p + 8 + self.L:p + 8 + self.L + osize]
            if idx == 0:
                break
            p += 8 + self.L + _pad8(osize)
        raise H5Error(f"global heap object {index} not found in the collection at {coll:#x}")

    @_guard
    def _attribute(self, off):
        buf = self.buf
        version = buf[off]
        nsize, tsize, ssize = struct.unpack_from("<HHH", buf, off + 2)
        if version == 1:
            p = off + 8
            name = buf[p:p + nsize].split(b"\x00")[0].decode("utf-8")
            p += _pad8(nsize)
            toff = p
            p += _pad8(tsize)
            soff = p
            p += _pad8(ssize)
        elif version in (2, 3):
            p = off + 8 + (1 if version == 3 else 0)
            name = buf[p:p + nsize].split(b"\x00")[0].decode("utf-8")
            p += nsize
            toff = p
            p += tsize
            soff = p
            p += ssize
        else:
            raise H5Error(f"attribute message version {version}")
        typ = _Type(buf, toff)
        shape = _dataspace(buf, soff, self.L)
        n = int(np.prod(shape)) if shape else 1
        if typ.cls == 9:
            if not typ.is_vlen_string:
                raise H5Error("variable-length sequences are not supported")
            vals = []
            for i in range(n):
                q = p + i * (4 + self.O + 4)
                (coll,) = self._unpack_O(q + 4)
                (idx,) = struct.unpack_from("<I", buf, q + 4 + self.O)
                vals.append(bytes(self._global_heap_object(coll, idx)))
            value = vals[0] if shape == () else np.array(vals, dtype=object).reshape(shape)
        else:
            arr = np.frombuffer(buf, dtype=typ.dtype, count=n, offset=p).reshape(shape).copy()
            value = arr[()] if shape == () else arr
            if typ.cls == 3 and shape == ():
                value = bytes(value)
        return name, value
