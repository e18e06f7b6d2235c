"""LMDB data-file reader (and a minimal bulk writer for tests) in pure Python.

The reference opens its sample cache with `lmdb.open(preloaded_dir, readonly=True, lock=False)` and fetches
`txn.get("{:010}".format(idx).encode("ascii"))` (data_loader/lmdb_data_loader.py:598-606, :631-634); `lmdb` is not in this
image and has no pure-Python build, so the on-disk format (LMDB 0.9, `data.mdb`; layout from the public mdb.c: MDB_page /
MDB_node / MDB_meta / MDB_db) is parsed directly:

  page   : u64 pgno | u16 pad | u16 flags | u16 lower | u16 upper  (overflow pages: u32 page count in place of lower/upper),
           then u16 node offsets [ (lower - 16) / 2 ], nodes packed from `upper` to the end of the page
  node   : u16 lo | u16 hi | u16 flags | u16 ksize | key | data      leaf: data size = lo | hi << 16, F_BIGDATA (0x01): the
           data field is the u64 number of the first overflow page;  branch: child page = lo | hi << 16 | flags << 32
  meta   : pages 0 and 1: u32 magic 0xBEEFC0DE | u32 version | u64 address | u64 mapsize | MDB_db free | MDB_db main |
           u64 last_pg | u64 txnid   (the copy with the larger txnid is current; free.md_pad holds the page size)
  MDB_db : u32 pad | u16 flags | u16 depth | u64 branch_pages | u64 leaf_pages | u64 overflow_pages | u64 entries | u64 root

Only what the reference uses is implemented: the unnamed main database, default (bytewise) key order, no DUPSORT.
NOTE (parity): no LMDB file written by liblmdb is available in the build container; this reader is tested against files
produced by `write_lmdb` below (same published layout) -- a file written by a reference environment should be added to
tests/golden/ when one is at hand."""
from __future__ import annotations

import mmap
import os
import struct
from typing import Dict, Iterator, Optional, Tuple

MAGIC, VERSION = 0xBEEFC0DE, 1
P_BRANCH, P_LEAF, P_OVERFLOW, P_META = 0x01, 0x02, 0x04, 0x08
F_BIGDATA = 0x01
PAGEHDR, NODEHDR = 16, 8
P_INVALID = 0xFFFFFFFFFFFFFFFF
_DB = struct.Struct("<IHHQQQQQ")          # MDB_db, 48 bytes
_META = struct.Struct("<IIQQ")            # magic, version, address, mapsize


class LMDBReader:
    def __init__(self, path: str):
        self.path = os.path.join(path, "data.mdb") if os.path.isdir(path) else path
        self._f = open(self.path, "rb")
        self._m = mmap.mmap(self._f.fileno(), 0, access=mmap.ACCESS_READ)
        metas = []
        for off in (0, None):
            if off is None:
                if not metas:
                    break
                off = metas[0]["psize"]
            if off + PAGEHDR + _META.size + 2 * _DB.size + 16 > len(self._m):
                continue
            magic, version, _addr, _mapsize = _META.unpack_from(self._m, off + PAGEHDR)
            if magic != MAGIC:
                if not metas:
                    raise ValueError(f"{self.path}: not an LMDB data file (magic {magic:#x})")
                continue
            free = _DB.unpack_from(self._m, off + PAGEHDR + _META.size)
            main = _DB.unpack_from(self._m, off + PAGEHDR + _META.size + _DB.size)
            last_pg, txnid = struct.unpack_from("<QQ", self._m, off + PAGEHDR + _META.size + 2 * _DB.size)
            metas.append(dict(version=version, psize=free[0], main=main, txnid=txnid, last_pg=last_pg))
        if not metas:
            raise ValueError(f"{self.path}: no valid meta page")
        meta = max(metas, key=lambda m: m["txnid"])
        if meta["version"] != VERSION:
            raise ValueError(f"{self.path}: unsupported LMDB data version {meta['version']}")
        self.psize = meta["psize"]
        _pad, self.flags, self.depth, self.branch_pages, self.leaf_pages, self.overflow_pages, self.entries, self.root = meta["main"]

    def close(self):
        self._m.close()
        self._f.close()

    def __len__(self) -> int:
        return self.entries

    def stat(self) -> Dict[str, int]:
        """like lmdb.Transaction.stat()"""
        return dict(psize=self.psize, depth=self.depth, branch_pages=self.branch_pages, leaf_pages=self.leaf_pages,
                    overflow_pages=self.overflow_pages, entries=self.entries)

    # ---- page helpers
    def _page(self, pgno: int) -> Tuple[int, int, int]:
        off = pgno * self.psize
        _pg, _pad, flags, lower, _upper = struct.unpack_from("<QHHHH", self._m, off)
        return off, flags, (lower - PAGEHDR) // 2

    def _node(self, page_off: int, i: int):
        (ptr,) = struct.unpack_from("<H", self._m, page_off + PAGEHDR + 2 * i)
        lo, hi, nflags, ksize = struct.unpack_from("<HHHH", self._m, page_off + ptr)
        return page_off + ptr, lo, hi, nflags, ksize

    def _key(self, node_off: int, ksize: int) -> bytes:
        return bytes(self._m[node_off + NODEHDR: node_off + NODEHDR + ksize])

    def _leaf_value(self, node_off: int, lo: int, hi: int, nflags: int, ksize: int) -> bytes:
        if nflags & ~F_BIGDATA:
            # F_SUBDATA (0x02) / F_DUPDATA (0x04): sub-databases and sorted duplicates -- not used by the reference's caches
            # (plain put() of one value per key); anything unknown is refused rather than mis-decoded
            raise ValueError(f"{self.path}: unsupported LMDB node flags {nflags:#x} (only plain values and F_BIGDATA are read)")
        size = lo | (hi << 16)
        d = node_off + NODEHDR + ksize
        if nflags & F_BIGDATA:
            (pgno,) = struct.unpack_from("<Q", self._m, d)
            start = pgno * self.psize + PAGEHDR
            return bytes(self._m[start:start + size])
        return bytes(self._m[d:d + size])

    def get(self, key: bytes) -> Optional[bytes]:
        if self.root == P_INVALID or self.entries == 0:
            return None
        pgno = self.root
        while True:
            off, flags, n = self._page(pgno)
            if flags & P_BRANCH:
                lo_i, hi_i = 1, n - 1          # node 0 carries no key: it covers everything below node 1's key
                child = 0
                while lo_i <= hi_i:            # last node whose key <= key
                    mid = (lo_i + hi_i) // 2
                    node_off, _l, _h, _f, ks = self._node(off, mid)
                    if self._key(node_off, ks) <= key:
                        child, lo_i = mid, mid + 1
                    else:
                        hi_i = mid - 1
                _no, l, h, f, _ks = self._node(off, child)
                pgno = l | (h << 16) | (f << 32)
            elif flags & P_LEAF:
                lo_i, hi_i = 0, n - 1
                while lo_i <= hi_i:
                    mid = (lo_i + hi_i) // 2
                    node_off, l, h, f, ks = self._node(off, mid)
                    k = self._key(node_off, ks)
                    if k == key:
                        return self._leaf_value(node_off, l, h, f, ks)
                    if k < key:
                        lo_i = mid + 1
                    else:
                        hi_i = mid - 1
                return None
            else:
                raise ValueError(f"{self.path}: unexpected page flags {flags:#x} at page {pgno}")

    def items(self) -> Iterator[Tuple[bytes, bytes]]:
        if self.root == P_INVALID or self.entries == 0:
            return
        stack = [self.root]
        while stack:
            pgno = stack.pop()
            off, flags, n = self._page(pgno)
            if flags & P_BRANCH:
                kids = []
                for i in range(n):
                    _no, l, h, f, _ks = self._node(off, i)
                    kids.append(l | (h << 16) | (f << 32))
                stack.extend(reversed(kids))
            else:
                for i in range(n):
                    node_off, l, h, f, ks = self._node(off, i)
                    yield self._key(node_off, ks), self._leaf_value(node_off, l, h, f, ks)


# ---------------------------------------------------------------------------------------------------------------------
def write_lmdb(path: str, items: Dict[bytes, bytes], psize: int = 4096) -> None:
    """Bulk-write `items` as a fresh LMDB environment directory (data.mdb + empty lock.mdb): sorted leaves filled left to
    right, values above LMDB's node-size limit on overflow pages, branch levels on top.  For tests and synthetic caches."""
    os.makedirs(path, exist_ok=True)
    nodemax = (((psize - PAGEHDR) // 2) & ~1) - 2
    pages: Dict[int, bytes] = {}
    next_pg = 2
    n_leaf = n_branch = n_ovf = 0

    def new_page(flags: int, nodes) -> int:
        """nodes: list of (hdr bytes (8), key, data); returns the page number"""
        nonlocal next_pg
        pgno = next_pg
        next_pg += 1
        buf = bytearray(psize)
        upper = psize
        ptrs = []
        for hdr, key, data in nodes:
            body = hdr + key + data
            size = (len(body) + 1) & ~1
            upper -= size
            buf[upper:upper + len(body)] = body
            ptrs.append(upper)
        lower = PAGEHDR + 2 * len(ptrs)
        assert lower <= upper, "page overflow"
        struct.pack_into("<QHHHH", buf, 0, pgno, 0, flags, lower, upper)
        for i, p in enumerate(ptrs):
            struct.pack_into("<H", buf, PAGEHDR + 2 * i, p)
        pages[pgno] = bytes(buf)
        return pgno

    def node_size(key, data_len):
        return ((NODEHDR + len(key) + data_len + 1) & ~1) + 2

    # leaves
    level = []           # (first key, pgno)
    cur, cur_used, first = [], PAGEHDR, None
    for key in sorted(items):
        val = items[key]
        if NODEHDR + len(key) + len(val) > nodemax:
            npg = (PAGEHDR + len(val) + psize - 1) // psize
            pg0 = next_pg
            next_pg += npg
            n_ovf += npg
            blob = bytearray(npg * psize)
            struct.pack_into("<QHHI", blob, 0, pg0, 0, P_OVERFLOW, npg)
            blob[PAGEHDR:PAGEHDR + len(val)] = val
            for j in range(npg):
                pages[pg0 + j] = bytes(blob[j * psize:(j + 1) * psize])
            hdr = struct.pack("<HHHH", len(val) & 0xFFFF, len(val) >> 16, F_BIGDATA, len(key))
            data = struct.pack("<Q", pg0)
        else:
            hdr = struct.pack("<HHHH", len(val) & 0xFFFF, len(val) >> 16, 0, len(key))
            data = val
        need = node_size(key, len(data))
        if cur and cur_used + need > psize:
            level.append((first, new_page(P_LEAF, cur)))
            n_leaf += 1
            cur, cur_used, first = [], PAGEHDR, None
        if first is None:
            first = key
        cur.append((hdr, key, data))
        cur_used += need
    if cur:
        level.append((first, new_page(P_LEAF, cur)))
        n_leaf += 1
    depth = 1 if level else 0
    while len(level) > 1:
        nxt, cur, cur_used, first = [], [], PAGEHDR, None
        for k, pg in level:
            key = b"" if not cur else k
            hdr = struct.pack("<HHHH", pg & 0xFFFF, (pg >> 16) & 0xFFFF, (pg >> 32) & 0xFFFF, len(key))
            need = node_size(key, 0)
            if cur and cur_used + need > psize:
                nxt.append((first, new_page(P_BRANCH, cur)))
                n_branch += 1
                cur, cur_used, first = [], PAGEHDR, None
                key = b""
                hdr = struct.pack("<HHHH", pg & 0xFFFF, (pg >> 16) & 0xFFFF, (pg >> 32) & 0xFFFF, 0)
                need = node_size(key, 0)
            if first is None:
                first = k
            cur.append((hdr, key, b""))
            cur_used += need
        nxt.append((first, new_page(P_BRANCH, cur)))
        n_branch += 1
        level = nxt
        depth += 1
    root = level[0][1] if level else P_INVALID
    last_pg = next_pg - 1
    mapsize = max(next_pg * psize, 1 << 20)
    for m, txnid in ((0, 0), (1, 1)):
        buf = bytearray(psize)
        struct.pack_into("<QHHHH", buf, 0, m, 0, P_META, 0, 0)
        _META.pack_into(buf, PAGEHDR, MAGIC, VERSION, 0, mapsize)
        _DB.pack_into(buf, PAGEHDR + _META.size, psize, 0, 0, 0, 0, 0, 0, P_INVALID)
        _DB.pack_into(buf, PAGEHDR + _META.size + _DB.size, 0, 0, depth, n_branch, n_leaf, n_ovf, len(items), root)
        struct.pack_into("<QQ", buf, PAGEHDR + _META.size + 2 * _DB.size, last_pg, txnid)
        pages[m] = bytes(buf)
    with open(os.path.join(path, "data.mdb"), "wb") as f:
        for pg in range(next_pg):
            f.write(pages[pg])
    open(os.path.join(path, "lock.mdb"), "wb").close()
