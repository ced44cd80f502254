"""Minimal pure-Python HDF5 reader / writer for Keras weight files (``model.save_weights("x.h5")``).

h5py is not importable by the interpreter this package runs on, and the reference's ``.h5`` files only use a
small corner of HDF5: the *old-style* group structure that libhdf5 writes by default (``libver='earliest'``,
superblock version 0, symbol-table groups = v1 B-tree + local heap + symbol nodes), version-1 object headers,
contiguous / compact little-endian datasets, and string attributes (fixed length or variable length).
That corner is what this module implements -- nothing else (no chunking, filters, new-style groups, references).

Format reference: "HDF5 File Format Specification Version 1.1/2.0" (The HDF Group, public).  Verified here against
files written by h5py 3.3.0 / libhdf5 1.12 (``tests/golden/make_h5_fixture.py`` wrote ``keras_like_*.h5``) and, the
other way round, by opening this writer's output with h5py (``tests/test_h5_cpu.py``, when an h5py interpreter exists).
"""
import struct

import numpy as np

SIG = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class HDF5Error(ValueError):
    pass


# =========================================================================================
# Reader
# =========================================================================================
class _Node(object):
    """A group or dataset: ``attrs`` dict, ``children`` (groups) or ``value`` (datasets)."""

    def __init__(self, name):
        self.name = name
        self.attrs = {}
        self.children = None      # OrderedDict-like dict name -> _Node for groups
        self.value = None         # numpy array for datasets

    @property
    def is_group(self):
        return self.children is not None

    def __getitem__(self, path):
        node = self
        for part in [p for p in path.split("/") if p]:
            if not node.is_group or part not in node.children:
                raise KeyError(path)
            node = node.children[part]
        return node

    def __contains__(self, path):
        try:
            self[path]
            return True
        except KeyError:
            return False


class _Reader(object):
    def __init__(self, buf):
        self.b = buf
        if buf[:8] != SIG:
            raise HDF5Error("not an HDF5 file (signature at offset 0 missing)")
        ver = buf[8]
        if ver not in (0, 1):
            raise HDF5Error(f"superblock version {ver} not supported (file not written with libver='earliest')")
        self.so, self.sl = buf[13], buf[14]
        if (self.so, self.sl) != (8, 8):
            raise HDF5Error("only 8-byte offsets / lengths are supported")
        p = 24 + (4 if ver == 1 else 0)
        self.base, _free, self.eof, _drv = struct.unpack_from("<QQQQ", buf, p)
        p += 32
        # root group symbol table entry
        _name_off, self.root_header, cache, _res = struct.unpack_from("<QQII", buf, p)
        self.gheaps = {}

    # ---- primitives ----
    def _local_heap_data(self, addr):
        if self.b[addr:addr + 4] != b"HEAP":
            raise HDF5Error("local heap signature missing")
        _size, _free, data_addr = struct.unpack_from("<QQQ", self.b, addr + 8)
        return data_addr

    def _cstr(self, addr):
        end = self.b.index(b"\x00", addr)
        return self.b[addr:end].decode("utf8")

    def _global_heap_object(self, coll_addr, index):
        if coll_addr not in self.gheaps:
            b = self.b
            if b[coll_addr:coll_addr + 4] != b"GCOL":
                raise HDF5Error("global heap collection signature missing")
            size = struct.unpack_from("<Q", b, coll_addr + 8)[0]
            objs, p, end = {}, coll_addr + 16, coll_addr + size
            while p + 16 <= end:
                idx, _ref, _r, osz = struct.unpack_from("<HHIQ", b, p)
                if idx == 0:
                    break
                objs[idx] = b[p + 16:p + 16 + osz]
                p += 16 + ((osz + 7) // 8) * 8
            self.gheaps[coll_addr] = objs
        return self.gheaps[coll_addr][index]

    # ---- messages ----
    def _messages(self, addr):
        b = self.b
        ver = b[addr]
        if ver != 1:
            raise HDF5Error(f"object header version {ver} not supported (new-style object header)")
        nmsg, _refc, hsize = struct.unpack_from("<HII", b, addr + 2)
        blocks = [(addr + 16, hsize)]
        out = []
        while blocks and len(out) < nmsg:
            p, size = blocks.pop(0)
            end = p + size
            while p + 8 <= end and len(out) < nmsg:
                mtype, msize, _flags = struct.unpack_from("<HHB", b, p)
                data = p + 8
                if mtype == 0x0010:
                    off, ln = struct.unpack_from("<QQ", b, data)
                    blocks.append((off, ln))
                out.append((mtype, data, msize))
                p = data + msize
        return out

    def _dataspace(self, p):
        ver, rank, flags = self.b[p], self.b[p + 1], self.b[p + 2]
        if ver == 1:
            q = p + 8
        elif ver == 2:
            if self.b[p + 3] == 2:       # null dataspace
                return None
            q = p + 4
        else:
            raise HDF5Error(f"dataspace version {ver}")
        return tuple(struct.unpack_from("<" + "Q" * rank, self.b, q)) if rank else ()

    def _datatype(self, p):
        """-> (kind, size, extra); kind in {'float', 'int', 'uint', 'str', 'vstr'}; returns also encoded length."""
        b = self.b
        cls, ver = b[p] & 0x0F, b[p] >> 4
        bits = b[p + 1] | (b[p + 2] << 8) | (b[p + 3] << 16)
        size = struct.unpack_from("<I", b, p + 4)[0]
        if cls == 0:       # fixed point
            if bits & 1:
                raise HDF5Error("big-endian integers not supported")
            return ("int" if (bits >> 3) & 1 else "uint", size, None), 8 + 4
        if cls == 1:       # floating point
            if bits & 1:
                raise HDF5Error("big-endian floats not supported")
            return ("float", size, None), 8 + 12
        if cls == 3:       # fixed-length string
            return ("str", size, None), 8
        if cls == 9:       # variable length
            if (bits & 0x0F) != 1:
                raise HDF5Error("variable-length sequences are not supported (only strings)")
            return ("vstr", size, None), 8 + 8 + 4
        raise HDF5Error(f"datatype class {cls} not supported")

    def _decode(self, dtype, shape, raw):
        kind, size, _ = dtype
        n = int(np.prod(shape)) if shape else 1
        if kind in ("float", "int", "uint"):
            code = {"float": "f", "int": "i", "uint": "u"}[kind]
            a = np.frombuffer(raw, dtype=np.dtype(f"<{code}{size}"), count=n)
            return a.reshape(shape).copy() if shape != () else a.reshape(()).copy()
        if kind == "str":
            vals = [bytes(raw[i * size:(i + 1) * size]).split(b"\x00")[0] for i in range(n)]
        else:
            vals = []
            for i in range(n):
                ln, coll, idx = struct.unpack_from("<IQI", raw, i * 16)
                vals.append(bytes(self._global_heap_object(coll, idx)[:ln]) if ln else b"")
        if shape == ():
            return vals[0]
        return np.array(vals, dtype=object).reshape(shape)

    def _attribute(self, p):
        b = self.b
        ver = b[p]
        if ver == 1:
            nsz, tsz, ssz = struct.unpack_from("<HHH", b, p + 2)
            q = p + 8
            pad = lambda v: (v + 7) // 8 * 8
        elif ver in (2, 3):
            nsz, tsz, ssz = struct.unpack_from("<HHH", b, p + 2)
            q = p + 8 + (1 if ver == 3 else 0)
            pad = lambda v: v
        else:
            raise HDF5Error(f"attribute message version {ver}")
        name = bytes(b[q:q + nsz]).split(b"\x00")[0].decode("utf8")
        q += pad(nsz)
        dtype, _ = self._datatype(q)
        q += pad(tsz)
        shape = self._dataspace(q)
        q += pad(ssz)
        if shape is None:
            return name, None
        n = int(np.prod(shape)) if shape else 1
        esize = 16 if dtype[0] == "vstr" else dtype[1]
        return name, self._decode(dtype, shape, b[q:q + n * esize])

    # ---- objects ----
    def _group_entries(self, btree, heap):
        """-> [(name, object header address)] in B-tree order."""
        b = self.b
        data = self._local_heap_data(heap)
        out = []

        def walk(addr):
            if b[addr:addr + 4] != b"TREE":
                raise HDF5Error("B-tree node signature missing")
            ntype, level, used = struct.unpack_from("<BBH", b, addr + 4)
            if ntype != 0:
                raise HDF5Error("not a group B-tree")
            p = addr + 24
            children = []
            for i in range(used):
                p += 8                                   # key i
                children.append(struct.unpack_from("<Q", b, p)[0])
                p += 8
            for c in children:
                if level > 0:
                    walk(c)
                else:
                    if b[c:c + 4] != b"SNOD":
                        raise HDF5Error("symbol node signature missing")
                    nsym = struct.unpack_from("<H", b, c + 6)[0]
                    for k in range(nsym):
                        name_off, hdr = struct.unpack_from("<QQ", b, c + 8 + 40 * k)
                        out.append((self._cstr(data + name_off), hdr))
        walk(btree)
        return out

    def read_object(self, header_addr, name):
        node = _Node(name)
        dtype = shape = layout = None
        stab = None
        for mtype, p, msize in self._messages(header_addr):
            if mtype == 0x0011:
                stab = struct.unpack_from("<QQ", self.b, p)
            elif mtype == 0x0001:
                shape = self._dataspace(p)
            elif mtype == 0x0003:
                dtype, _ = self._datatype(p)
            elif mtype == 0x0008:
                layout = p
            elif mtype == 0x000C:
                k, v = self._attribute(p)
                node.attrs[k] = v
            elif mtype in (0x0002, 0x0006):
                raise HDF5Error("new-style (link message) groups are not supported")
        if stab is not None:
            node.children = {}
            for cname, hdr in self._group_entries(*stab):
                node.children[cname] = self.read_object(hdr, cname)
            return node
        if dtype is None or layout is None:
            raise HDF5Error(f"object {name!r} is neither a symbol-table group nor a dataset")
        b = self.b
        ver, cls = b[layout], b[layout + 1]
        if ver != 3:
            raise HDF5Error(f"data layout version {ver} not supported")
        n = int(np.prod(shape)) if shape else 1
        esize = 16 if dtype[0] == "vstr" else dtype[1]
        if cls == 0:
            size = struct.unpack_from("<H", b, layout + 2)[0]
            raw = b[layout + 4:layout + 4 + size]
        elif cls == 1:
            addr, size = struct.unpack_from("<QQ", b, layout + 2)
            raw = b"\x00" * (n * esize) if addr == UNDEF else b[addr:addr + size]
        else:
            raise HDF5Error("chunked datasets are not supported (Keras weight files are contiguous)")
        node.value = self._decode(dtype, shape if shape is not None else (), raw)
        return node


def read_hdf5(path):
    """Parse a (Keras-weights style) HDF5 file into a tree of nodes with ``attrs`` / ``children`` / ``value``."""
    with open(path, "rb") as fh:
        buf = fh.read()
    r = _Reader(buf)
    return r.read_object(r.root_header, "/")


# =========================================================================================
# Writer
# =========================================================================================
class _Writer(object):
    """Writes old-style groups whose symbol tables fit one symbol node (<= 2 * LEAF_K entries per group)."""
    LEAF_K, INTERNAL_K = 32, 16

    def __init__(self):
        self.buf = bytearray()

    def alloc(self, n, align=8):
        pad = (-len(self.buf)) % align
        self.buf += b"\x00" * pad
        addr = len(self.buf)
        self.buf += b"\x00" * n
        return addr

    def put(self, addr, data):
        self.buf[addr:addr + len(data)] = data

    # ---- message encoders ----
    @staticmethod
    def _msg(mtype, payload, flags=0):
        payload = payload + b"\x00" * ((-len(payload)) % 8)
        return struct.pack("<HHBBBB", mtype, len(payload), flags, 0, 0, 0) + payload

    @staticmethod
    def _dataspace(shape):
        return struct.pack("<BBBBI", 1, len(shape), 0, 0, 0) + b"".join(struct.pack("<Q", d) for d in shape)

    @staticmethod
    def _dtype_float(size):
        props = {4: struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127), 8: struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)}[size]
        # class 1 version 1; bit field: little endian, pad 0, mantissa normalisation 2 (implied msb), sign location
        bits0 = 0x20
        sign = {4: 31, 8: 63}[size]
        return struct.pack("<BBBBI", 0x11, bits0, sign, 0, size) + props

    @staticmethod
    def _dtype_int(size, signed):
        return struct.pack("<BBBBI", 0x10, 0x08 if signed else 0x00, 0, 0, size) + struct.pack("<HH", 0, size * 8)

    @staticmethod
    def _dtype_str(size):
        return struct.pack("<BBBBI", 0x13, 0x00, 0, 0, size)       # null terminated / padded, ASCII

    def _encode_array(self, a):
        """-> (datatype bytes, shape, raw little-endian bytes)"""
        if isinstance(a, (bytes, str)):
            a = np.array(a.encode("utf8") if isinstance(a, str) else a)
        a = np.asarray(a)
        if a.dtype.kind in ("S", "O", "U"):
            items = [x.encode("utf8") if isinstance(x, str) else bytes(x) for x in a.reshape(-1).tolist()] if a.shape else \
                [a.item().encode("utf8") if isinstance(a.item(), str) else bytes(a.item())]
            size = max([len(x) for x in items] + [1])
            raw = b"".join(x.ljust(size, b"\x00") for x in items)
            return self._dtype_str(size), tuple(a.shape), raw
        if a.dtype.kind == "f":
            a = a.astype("<f4" if a.dtype.itemsize == 4 else "<f8")
            return self._dtype_float(a.dtype.itemsize), tuple(a.shape), a.tobytes()
        if a.dtype.kind in ("i", "u"):
            a = a.astype(a.dtype.newbyteorder("<"))
            return self._dtype_int(a.dtype.itemsize, a.dtype.kind == "i"), tuple(a.shape), a.tobytes()
        raise HDF5Error(f"cannot store dtype {a.dtype}")

    def _attr_msg(self, name, value):
        dt, shape, raw = self._encode_array(value)
        nm = name.encode("utf8") + b"\x00"
        ds = self._dataspace(shape)
        pad = lambda x: x + b"\x00" * ((-len(x)) % 8)
        body = struct.pack("<BBHHH", 1, 0, len(nm), len(dt), len(ds)) + pad(nm) + pad(dt) + pad(ds) + raw
        return self._msg(0x000C, body)

    def _object_header(self, msgs):
        body = b"".join(msgs)
        addr = self.alloc(16 + len(body))
        self.put(addr, struct.pack("<BBHII", 1, 0, len(msgs), 1, len(body)) + b"\x00" * 4 + body)
        return addr

    # ---- objects ----
    def write_dataset(self, array, attrs):
        dt, shape, raw = self._encode_array(array)
        data_addr = self.alloc(max(len(raw), 1))
        self.put(data_addr, raw)
        layout = struct.pack("<BBQQ", 3, 1, data_addr, len(raw))
        fill = struct.pack("<BBBB", 2, 2, 0, 0)         # fill value v2: allocate late, write never... "undefined"
        msgs = [self._msg(0x0001, self._dataspace(shape)), self._msg(0x0003, dt, flags=1), self._msg(0x0005, fill),
                self._msg(0x0008, layout)]
        msgs += [self._attr_msg(k, v) for k, v in attrs.items()]
        return self._object_header(msgs)

    def write_group(self, children, attrs):
        """children: {name: header address}"""
        names = sorted(children)                        # symbol nodes are ordered by name
        if len(names) > 2 * self.LEAF_K:
            raise HDF5Error(f"group with {len(names)} entries exceeds this writer's single symbol node")
        # local heap: offset 0 = empty string (left-most B-tree key), then the names, 8-byte aligned
        heap = bytearray(b"\x00" * 8)
        offs = {}
        for n in names:
            offs[n] = len(heap)
            e = n.encode("utf8") + b"\x00"
            heap += e + b"\x00" * ((-len(e)) % 8)
        free_off = len(heap)
        heap += struct.pack("<QQ", 1, 32) + b"\x00" * 16       # one free block (next = 1 = none, size 32) so the list is valid
        heap_data = self.alloc(len(heap))
        self.put(heap_data, bytes(heap))
        heap_addr = self.alloc(32)
        self.put(heap_addr, b"HEAP" + struct.pack("<BBBBQQQ", 0, 0, 0, 0, len(heap), free_off, heap_data))
        # symbol node
        snod = self.alloc(8 + 2 * self.LEAF_K * 40)
        body = b"SNOD" + struct.pack("<BBH", 1, 0, len(names))
        for n in names:
            body += struct.pack("<QQII", offs[n], children[n], 0, 0) + b"\x00" * 16
        self.put(snod, body)
        # B-tree: one leaf-level node with one child
        tree = self.alloc(24 + (2 * self.INTERNAL_K + 1) * 8 + 2 * self.INTERNAL_K * 8)
        last = offs[names[-1]] if names else 0
        self.put(tree, b"TREE" + struct.pack("<BBHQQ", 0, 0, 1 if names else 0, UNDEF, UNDEF) + struct.pack("<QQQ", 0, snod, last))
        msgs = [self._msg(0x0011, struct.pack("<QQ", tree, heap_addr))] + [self._attr_msg(k, v) for k, v in attrs.items()]
        return self._object_header(msgs), tree, heap_addr

    def finish(self, root_header, root_tree, root_heap):
        sb = SIG + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, self.LEAF_K, self.INTERNAL_K, 0)
        sb += struct.pack("<QQQQ", 0, UNDEF, len(self.buf), UNDEF)
        sb += struct.pack("<QQII", 0, root_header, 1, 0) + struct.pack("<QQ", root_tree, root_heap)
        self.put(0, sb)


def write_hdf5(path, tree):
    """``tree``: nested dict ``{"attrs": {...}, "groups": {name: tree}, "datasets": {name: (array, attrs)}}``."""
    w = _Writer()
    w.alloc(96)                       # superblock (56 bytes + 40-byte root symbol table entry)

    def emit(t):
        children = {}
        for name, sub in t.get("groups", {}).items():
            children[name] = emit(sub)[0]
        for name, (arr, attrs) in t.get("datasets", {}).items():
            children[name] = w.write_dataset(arr, attrs or {})
        return w.write_group(children, t.get("attrs", {}))

    hdr, tr, hp = emit(tree)
    w.finish(hdr, tr, hp)
    with open(path, "wb") as fh:
        fh.write(bytes(w.buf))
