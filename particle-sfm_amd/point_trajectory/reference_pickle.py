"""track.npy in the REFERENCE's pickle state, streamed from the CSR arrays ("emit the identical pickle object
graph straight from SoA arrays").

The reference's `TrajectorySet` pickles as `{id: {"frame_ids": [int], "locations": [V2D], "labels": [bool]}}`
(optimize/src/bindings.cc:64-71 over Trajectory::as_dict, trajectory_base.cpp:47-53).  Built as Python objects that is 1.9 M dicts
and 5e7 list elements for BASELINE configs[1] (45 s to write, 50 s to read back in the build container).  Here the pickle
OPCODES are written directly, one fixed-size record per trajectory, filled in with NumPy:

    J id  }  (  "frame_ids"  list(range(b, b + n))
                "locations"  getitem(XY, slice(s, e))        XY = the (n_points, 2) f64 array, pickled ONCE in front
                "labels"     mul([False], n)
             u

`list`, `range`, `slice`, `operator.getitem`, `operator.mul`, the three keys, `[False]` and XY are memoised once and fetched with
BINGET; the outer dict takes all records in one SETITEMS.  What a consumer unpickles is the same mapping with the same values:
`frame_ids` a list of ints, `labels` a list of bools, `locations` an (n, 2) float64 array (a view of XY -- pybind's
`std::vector<V2D>` caster and `np.array()` take it row by row, exactly like the list of 2-vectors the reference's own dump holds).
The container around it (the .npy header, the 0-d object array, the class reference) is taken from a real `pickle.dumps` of a
template instance, so it is whatever this NumPy / this class produce.
"""
import pickle
import pickletools
import struct
import threading

import numpy as np

# Behind the pickle's STOP the writer leaves a 64-byte footer (np.load / pickle.load stop reading at STOP and never see it): where
# the point array's raw bytes and the records sit in the file.  `load()` uses it to read its own files back at array speed --
# the reference layout itself can only be unpickled one Python object per frame id and label (22 s for 1.9 M trajectories).
_FOOTER_MAGIC = b"PSFMTRK1"
_FOOTER = struct.Struct("<8sqqqqqq8s")        # magic, xy offset, n_points, records offset, n, record size, file size before the footer, magic

_SENTINEL = b"@@PSFM-TRAJECTORY-SET-STATE@@"
# memo slots of the preamble (the template uses single-digit slots)
_M_XY, _M_LIST, _M_RANGE, _M_SLICE, _M_GETITEM, _M_MUL, _M_FALSE, _M_KF, _M_KL, _M_KB = range(200, 210)


def _global(module, name, slot):
    return b"c" + module + b"\n" + name + b"\n" + b"q" + bytes([slot]) + b"0"        # GLOBAL, BINPUT slot, POP


def _key(text, slot):
    return b"X" + struct.pack("<I", len(text)) + text + b"q" + bytes([slot]) + b"0"   # BINUNICODE, BINPUT slot, POP


def _record_template():
    """One trajectory's opcodes with zeroed integer fields; returns (bytes, offsets of the 4-byte fields id, b, b+n, s, e, n)."""
    g = lambda slot: b"h" + bytes([slot])                                             # BINGET
    J = b"J\x00\x00\x00\x00"                                                          # BININT (4-byte signed)
    parts, at = [], {}

    def put(x, name=None):
        if name:
            at[name] = sum(len(q) for q in parts) + 1
        parts.append(x)
    put(J, "id"); put(b"}("); put(g(_M_KF))
    put(g(_M_LIST)); put(g(_M_RANGE)); put(J, "b"); put(J, "bn"); put(b"\x86R\x85R")       # list(range(b, b + n))
    put(g(_M_KL))
    put(g(_M_GETITEM)); put(g(_M_XY)); put(g(_M_SLICE)); put(J, "s"); put(J, "e"); put(b"\x86R\x86R")   # getitem(XY, slice(s, e))
    put(g(_M_KB))
    put(g(_M_MUL)); put(g(_M_FALSE)); put(J, "n"); put(b"\x86R")                          # mul([False], n)
    put(b"u")
    return b"".join(parts), at


def can_stream(ts):
    """The streaming form covers what the builder produces: a CSR-backed set without labels, fewer than 2^31 points."""
    csr = getattr(ts, "_csr", None)
    return (csr is not None and getattr(ts, "_map", 1) is None and csr[5] is None and int(csr[3][-1]) < (1 << 31)
            and (len(csr[0]) == 0 or (int(np.max(csr[0])) < (1 << 31) and int(np.min(csr[0])) >= 0)))


def _write_front(fp, prefix, xy):
    """Everything in front of the records: the container's prefix, the shared objects (the point array among them), the dict's MARK.
    Returns (file offset of the raw point bytes or -1, the point array as written, file offset of the first record)."""
    fp.write(prefix)
    # preamble: the shared objects, each memoised and popped again
    xy = np.ascontiguousarray(xy, dtype=np.float64).reshape(-1, 2)
    # XY as hand-written opcodes, no PROTO / FRAME inside the stream (a protocol-5 sub-pickle would leave its last FRAME open across
    # the opcodes that follow: the C unpickler tolerates that, `pickle._Unpickler` -- PyPy's only one -- does not):
    #     <what ndarray.__reduce_ex__(5) names, e.g. numpy._core.numeric._frombuffer>(BYTEARRAY8 raw, dtype('<f8'), (n, 2), 'C')
    # The raw bytes go from the array to the file without an intermediate copy; BYTEARRAY8 keeps the array writable.
    rebuild = np.zeros((1, 2)).__reduce_ex__(5)[0]
    item = lambda o: pickletools.optimize(pickle.dumps(o, protocol=3))[2:-1]     # protocol 3: no framing; optimize: no memo slots
    fp.write(b"c" + rebuild.__module__.encode() + b"\n" + rebuild.__name__.encode() + b"\n(")      # GLOBAL, MARK
    fp.write(b"\x96" + struct.pack("<Q", xy.nbytes))                                                 # BYTEARRAY8
    xy_data = fp.tell() if xy.nbytes > 0 else -1
    if xy.nbytes > 0:
        fp.write(xy.reshape(-1).view(np.uint8).data)
    fp.write(item(xy.dtype) + item(tuple(int(x) for x in xy.shape)) + item("C") + b"tR")             # ... TUPLE, REDUCE
    fp.write(b"q" + bytes([_M_XY]) + b"0")
    fp.write(_global(b"builtins", b"list", _M_LIST) + _global(b"builtins", b"range", _M_RANGE) + _global(b"builtins", b"slice", _M_SLICE)
             + _global(b"operator", b"getitem", _M_GETITEM) + _global(b"operator", b"mul", _M_MUL)
             + b"]\x89a" + b"q" + bytes([_M_FALSE]) + b"0"                      # EMPTY_LIST NEWFALSE APPEND -> [False]
             + _key(b"frame_ids", _M_KF) + _key(b"locations", _M_KL) + _key(b"labels", _M_KB))
    # the state: one dict, all records under one MARK ... SETITEMS
    fp.write(b"}(")
    return xy_data, xy, fp.tell()


def dump(fp, ts):
    """Write the object array holding `ts` (what np.save(path, ts) writes behind the .npy header) to the open binary file."""
    ids, birth, length, off, xy, _ = ts._csr
    n = len(ids)
    # the container: pickle a template instance whose state is a sentinel, cut the stream at the sentinel
    tmpl = type(ts).__new__(type(ts))
    tmpl.__dict__["_state_override"] = _SENTINEL
    arr = np.empty((), dtype=object)
    arr[()] = tmpl
    stream = pickle.dumps(arr, protocol=3)
    mark = b"C" + bytes([len(_SENTINEL)]) + _SENTINEL
    i = stream.index(mark)
    j = i + len(mark)
    assert stream[j:j + 1] == b"q" and stream.count(mark) == 1, "unexpected pickle layout of the template"
    prefix, suffix = stream[:i], stream[j + 2:]
    # The records (66 bytes per trajectory, filled in with NumPy) are built on a helper thread while this one writes the point array --
    # the bulk of the file, a single write() that spends its time in the kernel's page-cache copy with the GIL released.
    rec, at = _record_template()
    R = len(rec)
    step = 1 << 18
    ids = np.asarray(ids, np.int64); birth = np.asarray(birth, np.int64); length = np.asarray(length, np.int64); off = np.asarray(off, np.int64)
    # ... handed over through a BOUNDED queue (a few chunks of 256 k records = 17 MB each in flight, whatever the number of
    # trajectories) and written as they arrive once the point array is out; a writer that fails tells the builder to stop.
    import queue
    todo, failed, cancel = queue.Queue(maxsize=3), [], threading.Event()

    def build_records():
        try:
            for lo in range(0, n, step):
                if cancel.is_set():
                    return
                hi = min(n, lo + step)
                buf = np.tile(np.frombuffer(rec, np.uint8), hi - lo).reshape(hi - lo, R)
                fields = {"id": ids[lo:hi], "b": birth[lo:hi], "bn": birth[lo:hi] + length[lo:hi], "s": off[lo:hi], "e": off[lo + 1:hi + 1],
                          "n": length[lo:hi]}
                for name, v in fields.items():
                    buf[:, at[name]:at[name] + 4] = v.astype("<i4").view(np.uint8).reshape(-1, 4)
                while not cancel.is_set():
                    try:
                        todo.put(buf, timeout=0.1)
                        break
                    except queue.Full:
                        pass
        except BaseException as e:      # noqa: BLE001  (handed to the writing thread: a short record list must not reach the file)
            failed.append(e)
        finally:
            while not cancel.is_set():   # the end marker (behind a failure too: the writer must not wait for ever)
                try:
                    todo.put(None, timeout=0.1)
                    break
                except queue.Full:
                    pass
    builder = threading.Thread(target=build_records)
    builder.start()
    try:
        xy_data, xy, rec_start = _write_front(fp, prefix, xy)
        while True:
            buf = todo.get()
            if buf is None:
                break
            fp.write(memoryview(buf).cast("B"))
    finally:
        cancel.set()
        builder.join()
    if failed:
        raise failed[0]
    fp.write(b"u")
    fp.write(suffix)
    # footer: where the raw point bytes are
    end = fp.tell()
    if xy_data >= 0 or xy.nbytes == 0:
        fp.write(_FOOTER.pack(_FOOTER_MAGIC, xy_data, xy.shape[0], rec_start, n, R, end, _FOOTER_MAGIC))


def load(path):
    """track.npy -> TrajectorySet.  Files written by dump() are read back through their footer (CSR arrays straight from the
    file, nothing unpickled); anything else goes through np.load(path, allow_pickle=True).item()."""
    import os
    from .optimize.build import particlesfm
    try:
        size = os.path.getsize(path)
        with open(path, "rb") as fp:
            if size > _FOOTER.size:
                fp.seek(size - _FOOTER.size)
                m0, xy_data, n_pts, rec_start, n, R, end, m1 = _FOOTER.unpack(fp.read(_FOOTER.size))
                rec, at = _record_template()
                if (m0 == _FOOTER_MAGIC and m1 == _FOOTER_MAGIC and end == size - _FOOTER.size and R == len(rec) and n >= 0
                        and rec_start + n * R < end and (n_pts == 0 or 0 < xy_data < rec_start)):
                    fp.seek(rec_start)
                    buf = np.frombuffer(fp.read(n * R), np.uint8).reshape(n, R)
                    zero = np.frombuffer(rec, np.uint8).copy()
                    field = lambda name: np.ascontiguousarray(buf[:, at[name]:at[name] + 4]).view("<i4").reshape(-1).astype(np.int64)
                    ids, birth, bn, s0, e0, ln = (field(k) for k in ("id", "b", "bn", "s", "e", "n"))
                    chk = buf.copy()
                    for name in at:
                        chk[:, at[name]:at[name] + 4] = 0
                    off = np.zeros(n + 1, np.int64)
                    np.cumsum(ln, out=off[1:])
                    if (n == 0 or ((chk == zero).all() and np.array_equal(bn, birth + ln) and np.array_equal(s0, off[:-1])
                                   and np.array_equal(e0, off[1:]))) and int(off[-1]) == n_pts:
                        if n_pts:     # (mapped copy-on-write: pages come in as the consumer touches them)
                            xy = np.memmap(path, dtype=np.float64, mode="c", offset=xy_data, shape=(n_pts, 2))
                        else:
                            xy = np.zeros((0, 2))
                        return particlesfm.TrajectorySet._from_csr(ids, birth.astype(np.int32), ln.astype(np.int32), off, xy)
    except (OSError, ValueError, struct.error):
        pass
    return np.load(path, allow_pickle=True).item()
