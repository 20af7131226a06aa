"""RankingEvent -> mrk_request (include/mrk.h) marshalling for the ctypes binding.

Input is the JSON shape of Metarank's `ranking` event (doc/event-schema.md; decoder
M/model/Event.scala:83-97): {"id", "timestamp", "user", "session", "fields": [{"name","value"}],
"items": [{"id", "relevancy"?, "fields"?: [...]}]}.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._native import mrk_field, mrk_request

F_STRING, F_NUMBER, F_BOOL, F_STRING_LIST, F_NUMBER_LIST = 0, 1, 2, 3, 4


def _fill_field(dst: mrk_field, name: str, value, keep: list):
    nb = name.encode()
    keep.append(nb)
    dst.name = nb
    if isinstance(value, bool):
        dst.type, dst.num = F_BOOL, 1.0 if value else 0.0
    elif isinstance(value, (int, float)):
        dst.type, dst.num = F_NUMBER, float(value)
    elif isinstance(value, str):
        b = value.encode()
        keep.append(b)
        dst.type, dst.str = F_STRING, b
    elif isinstance(value, (list, tuple, np.ndarray)):
        if len(value) > 0 and isinstance(value[0], str):
            bs = [v.encode() for v in value]
            arr = (C.c_char_p * len(bs))(*bs)
            keep.extend([bs, arr])
            dst.type, dst.n, dst.strs = F_STRING_LIST, len(bs), arr
        else:
            a = np.ascontiguousarray(value, dtype=np.float64)
            keep.append(a)
            dst.type, dst.n = F_NUMBER_LIST, len(a)
            dst.nums = a.ctypes.data_as(C.POINTER(C.c_double))
    else:
        raise TypeError(f"field {name}: unsupported value {value!r}")


class Request:
    """Owns the ctypes memory of one mrk_request."""

    def __init__(self, event: dict, with_ids: bool = True):
        """with_ids False: item_ids stays NULL (the ids travel as flat bytes, RequestSet / mrk_item_ids)"""
        self.event = event
        self._keep: list = []
        c = mrk_request()
        k = self._keep

        def s(v):
            if v is None:
                return None
            b = str(v).encode()
            k.append(b)
            return b

        c.id = s(event.get("id", ""))
        c.timestamp_ms = int(event.get("timestamp", 0))
        c.user = s(event.get("user"))
        c.session = s(event.get("session"))
        fields = event.get("fields") or []
        farr = (mrk_field * max(len(fields), 1))()
        for i, f in enumerate(fields):
            _fill_field(farr[i], f["name"], f["value"], k)
        k.append(farr)
        c.fields, c.n_fields = farr, len(fields)
        items = event["items"]
        ids = [str(it["id"] if isinstance(it, dict) else it).encode() for it in items]
        self.id_bytes = ids
        if with_ids:
            idarr = (C.c_char_p * max(len(ids), 1))(*ids)
            k.extend([ids, idarr])
            c.item_ids = idarr
        c.n_items = len(ids)
        per_item = []
        for it in items:
            fl = []
            if isinstance(it, dict):
                if it.get("relevancy") is not None:  # RankItem decoder: relevancy first, then fields
                    fl.append({"name": "relevancy", "value": float(it["relevancy"])})
                fl.extend(it.get("fields") or [])
            per_item.append(fl)
        total = sum(len(fl) for fl in per_item)
        if total:
            offs = (C.c_int32 * (len(items) + 1))()
            ifarr = (mrk_field * total)()
            p = 0
            for i, fl in enumerate(per_item):
                offs[i] = p
                for f in fl:
                    _fill_field(ifarr[p], f["name"], f["value"], k)
                    p += 1
            offs[len(items)] = p
            k.extend([offs, ifarr])
            c.item_field_offsets, c.item_fields = offs, ifarr
        self.c = c

    @property
    def n_items(self) -> int:
        return self.c.n_items


def request_array(reqs):
    arr = (mrk_request * max(len(reqs), 1))()
    for i, r in enumerate(reqs):
        C.memmove(C.byref(arr, i * C.sizeof(mrk_request)), C.byref(r.c), C.sizeof(mrk_request))
    return arr


class RequestSet:
    """Requests marshalled for the serving loop (mrk_batch_load with mrk_item_ids): the mrk_request array without
    per-item C strings, and the UTF-8 bytes of all item ids back to back + their offsets - in pinned memory
    (mrk_host_alloc) unless pinned=False, so that the copy engine reads them where they are."""

    def __init__(self, events, pinned: bool = True):
        from . import _native as N

        self.requests = [e if isinstance(e, Request) else Request(e, with_ids=False) for e in events]
        self.arr = request_array(self.requests)
        ids = [b for r in self.requests for b in r.id_bytes]
        lens = np.fromiter((len(b) for b in ids), dtype=np.int64, count=len(ids))
        offs = np.zeros(len(ids) + 1, dtype=np.uint32)
        np.cumsum(lens, out=offs[1:])
        blob = b"".join(ids)
        self.total_items = len(ids)
        self.n_req = len(self.requests)
        self.offsets_per_request = np.concatenate([[0], np.cumsum([r.n_items for r in self.requests])]).astype(np.int64)
        self._pinned = []
        if pinned:
            def pin(nbytes):
                p = N.lib().mrk_host_alloc(max(nbytes, 1))
                if not p:
                    raise MemoryError("mrk_host_alloc failed")
                self._pinned.append(p)
                return p
            pb, po = pin(len(blob)), pin(offs.nbytes)
            C.memmove(pb, blob, len(blob))
            C.memmove(po, offs.ctypes.data, offs.nbytes)
            self._blob, self._offs = None, None
            self.ids = N.mrk_item_ids(pb, po, len(blob))
        else:
            self._blob = np.frombuffer(blob, dtype=np.uint8) if blob else np.zeros(1, dtype=np.uint8)
            self._offs = offs
            self.ids = N.mrk_item_ids(self._blob.ctypes.data, offs.ctypes.data, len(blob))
        self.id_bytes_total = len(blob)

    def requests_with_ids(self):
        """the same requests with per-item C strings (pointer-style mrk_batch_prepare), for cross-checks"""
        return [Request(r.event) for r in self.requests]

    def close(self):
        from . import _native as N

        for p in self._pinned:
            N.lib().mrk_host_free(p)
        self._pinned = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
