"""The drop-in boundary as a maintainer of the reference would use it: INTEGRATION.md section 2's ctypes stub is
EXTRACTED from the file and executed verbatim, and the C-ABI pipeline (mfar_pipeline_*) is driven through its Python face -- ids, score
bits and n_valid against MultiFieldIndex.search and the C oracle."""
import ctypes
import os
import re

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import mfar_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def integration_stub():
    """The python block of INTEGRATION.md section 2, as text."""
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = md[md.index("## 2. The binding a maintainer"):]
    m = re.search(r"```python\n(.*?)```", sec, re.S)
    assert m, "INTEGRATION.md section 2 holds no python block"
    return m.group(1)


def load_stub():
    """Execute the stub verbatim.  It opens the library by its soname: map the in-tree build first, so that the loader finds it."""
    from mfar import _native
    _native.lib()                                            # builds the error message when the library is missing; imports torch first
    ctypes.CDLL(_native.LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    ns = {"__name__": "mfar_native_stub"}
    exec(compile(integration_stub(), "INTEGRATION.md#2", "exec"), ns)
    return ns


def _mk(rng, F, D, E):
    mu = rng.standard_normal(E).astype(np.float32)
    mu /= np.linalg.norm(mu)
    slab = (rng.standard_normal((F, D, E)) * 0.5 + mu * 1.2).astype(np.float32)
    W = (rng.standard_normal((E, F)) * 0.05).astype(np.float32)
    return slab, mu, W


def test_integration_stub_runs_verbatim_and_matches_the_oracle():
    import torch
    ns = load_stub()
    rng = np.random.default_rng(5)
    F, D, E, Q = 4, 30000, 96, 64
    slab, mu, W = _mk(rng, F, D, E)
    qs = [(rng.standard_normal((Q if i != 4 else 23, E)) * 0.5 + mu * 2.0).astype(np.float32) for i in range(9)]
    mask = np.array([1, 1, 0, 1], np.float32)
    dev = torch.device("cuda:0")
    ix = ns["HbmIndex"](D, F, E, 0, 0)
    for f in range(F):
        ix.write(f, 0, torch.from_numpy(slab[f]).to(dev))
    Wd, md = torch.from_numpy(W).to(dev), torch.from_numpy(mask).to(dev)
    want = [O.c_two_stage(slab, q, W, mask) for q in qs[:3]]
    # the synchronous binding
    sync = [ix.search(torch.from_numpy(q).to(dev), Wd, md) for q in qs]
    torch.cuda.synchronize()
    for (ids, sc, nv), o in zip(sync, want):
        assert np.array_equal(ids.cpu().numpy(), o["ids"]) and np.array_equal(sc.cpu().numpy().view(np.uint32), o["scores"].view(np.uint32))
        assert (nv.cpu().numpy() == 100).all()
    # the pipelined binding: results `lag` submissions late, a short batch in the middle, the last ones drained
    pipe = ns["HbmPipeline"](ix, Wd, md)
    assert pipe.lag == 5                                       # three launches x two coalesced batches - 1
    tickets, got = [], []
    for i, q in enumerate(qs):
        tickets.append(pipe.submit(torch.from_numpy(q).to(dev)))
        if i >= pipe.lag:
            got.append(pipe.result(tickets[i - pipe.lag]))
    for t in tickets[len(got):]:
        got.append(pipe.result(t))
    torch.cuda.synchronize()
    for (ids, sc, nv), (ids0, sc0, nv0) in zip(got, sync):
        assert torch.equal(ids, ids0) and torch.equal(sc, sc0) and torch.equal(nv, nv0)
    pipe.close()
    ix.close()


def test_native_pipeline_equals_search(idxmod=None):
    """mfar_pipeline_* through mfar.data.pipeline.NativePipeline: device and host batches, coalescing on and off, every depth, a forced
    failure of every certificate (redone when the result is taken, then repaired inline), new weights in the middle -- always the bits of
    MultiFieldIndex.search."""
    import torch
    from mfar.data import index as idxmod
    from mfar.data.pipeline import NativePipeline
    rng = np.random.default_rng(6)
    F, D, E, Q = 3, 24000, 64, 48
    slab, mu, W = _mk(rng, F, D, E)
    W2 = (rng.standard_normal((E, F)) * 0.05).astype(np.float32)
    mask = np.array([1, 0, 1], np.float32)
    ix = idxmod.MultiFieldIndex(D, F, E, device=0)
    for f in range(F):
        ix.write_rows(f, 0, slab[f])
    dev = torch.device("cuda:0")
    qs = [(rng.standard_normal((Q if i % 4 else 17, E)) * 0.5 + mu * 2.0).astype(np.float32) for i in range(11)]
    ix.set_screen(0)
    want = [ix.search(q, W, mask) for q in qs]
    want2 = [ix.search(q, W2, None) for q in qs]
    for screen, eps_mult, depth, coalesce, on_dev in ((2, 1.0, 0, 0, True), (2, 1.0, 2, 1, False), (2, 1e9, 4, 0, True), (0, 1.0, 3, 0, False)):
        ix.set_screen(screen, eps_mult)
        Wd, md = (torch.from_numpy(W).to(dev), torch.from_numpy(mask).to(dev)) if on_dev else (W, mask)
        pl = NativePipeline(ix, Wd, md, max_batch=Q, depth=depth, coalesce=coalesce)
        assert pl.coalesce == (coalesce or (2 if screen else 1)) and pl.depth == (depth or 3) and pl.lag == pl.depth * pl.coalesce - 1
        tickets, got = [], []
        for i, q in enumerate(qs):
            tickets.append(pl.submit(torch.from_numpy(q).to(dev) if on_dev else q))
            if i >= pl.lag:
                got.append(pl.result(tickets[i - pl.lag]))
        for t in tickets[len(got):]:
            got.append(pl.result(t))
        with pytest.raises(Exception):
            pl.result(tickets[0])                              # long out of flight
        for g, w in zip(got, want):
            gi, gs, gn = ((x.cpu().numpy() if on_dev else x) for x in (g["ids"], g["scores"], g["n_valid"]))
            assert np.array_equal(gi, w["ids"]) and np.array_equal(gs.view(np.uint32), w["scores"].view(np.uint32)) and np.array_equal(gn, w["n_valid"])
        assert (pl.n_redone >= 1) == (eps_mult > 1.0), (pl.n_redone, eps_mult)
        # new weights, no mask: flushes, waits, replaces
        pl.set_weights(torch.from_numpy(W2).to(dev) if on_dev else W2, None)
        for q, w in zip(qs[:3], want2):
            g = pl.result(pl.submit(torch.from_numpy(q).to(dev) if on_dev else q))      # (taken at once: a held batch is launched alone)
            gi, gs = ((x.cpu().numpy() if on_dev else x) for x in (g["ids"], g["scores"]))
            assert np.array_equal(gi, w["ids"]) and np.array_equal(gs.view(np.uint32), w["scores"].view(np.uint32))
        pl.close()
    ix.close()


def test_plain_c_host_over_the_abi(tmp_path):
    """No Python, no torch on the host side: tests/host/abi_client.c (C11, gcc) includes include/mfar_hip.h, links libmfar_hip.so, builds an
    index from host memory and runs both the synchronous scorer and the batch pipeline on host buffers.  It checks pipeline == synchronous
    itself; here its outputs are compared with the C oracle bit for bit."""
    import shutil
    import subprocess
    from mfar import _native
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    _native.lib()
    exe, out = str(tmp_path / "abi_client"), str(tmp_path / "out.bin")
    libdir = os.path.dirname(_native.LIB_PATH)
    subprocess.check_call([gcc, "-std=c11", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "host", "abi_client.c"), "-L", libdir, "-lmfar_hip", f"-Wl,-rpath,{libdir}", "-o", exe])
    D, F, E, Q, NB = 20000, 3, 64, 40, 7
    r = subprocess.run([exe, out, str(D), str(F), str(E), str(Q), str(NB)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("OK "), (r.stdout, r.stderr[-2000:])
    assert "coalesce=2" in r.stdout and "queries_per_launch=80" in r.stdout, r.stdout      # 20 000 rows: screened, wide pass, two batches per launch
    raw = np.fromfile(out, dtype=np.uint8)
    hdr = raw[:20].view(np.int32)
    assert hdr.tolist() == [D, F, E, Q, NB]
    o = 20
    def take(n, dt):
        nonlocal o
        a = raw[o:o + n * np.dtype(dt).itemsize].view(dt)
        o += n * np.dtype(dt).itemsize
        return a
    slab = take(F * D * E, np.float32).reshape(F, D, E)
    W = take(E * F, np.float32).reshape(E, F)
    mask = take(F, np.float32)
    q = take(NB * Q * E, np.float32).reshape(NB, Q, E)
    ids = take(NB * Q * 100, np.int64).reshape(NB, Q, 100)
    sc = take(NB * Q * 100, np.float32).reshape(NB, Q, 100)
    for b in (0, NB - 1):
        ref = O.c_two_stage(slab, q[b], W, mask)
        assert np.array_equal(ids[b], ref["ids"]) and np.array_equal(sc[b].view(np.uint32), ref["scores"].view(np.uint32)), b


def test_abi_error_paths_on_the_device():
    """Invalid requests against a LIVE index come back as error codes with a message (no crash, no partial launch): the same error
    behaviour the Python mirror turns into exceptions."""
    import ctypes
    from mfar import _native
    from mfar.data import index as idxmod
    L = _native.lib()
    rng = np.random.default_rng(9)
    F, D, E = 3, 20000, 64
    ix = idxmod.MultiFieldIndex(D, F, E, device=0)
    for f in range(F):
        ix.write_rows(f, 0, rng.standard_normal((D, E)).astype(np.float32))
    h = ix._h
    q = np.ascontiguousarray(rng.standard_normal((64, E)).astype(np.float32))
    W = np.ascontiguousarray(rng.standard_normal((E, F)).astype(np.float32))
    ids, sc, nv = np.empty((64, 100), np.int64), np.empty((64, 100), np.float32), np.empty(64, np.int32)
    P = lambda a: a.ctypes.data
    two = lambda **kw: L.mfar_search_two_stage(h, kw.get("q", P(q)), kw.get("Q", 64), kw.get("W", P(W)), 1, None, kw.get("k1", 100), kw.get("k2", 100), 1,
                                               kw.get("ids", P(ids)), P(sc), P(nv), None, None, None, 0, None)
    assert two() == 0
    for kw in (dict(k1=0), dict(k1=129), dict(k2=0), dict(k2=129), dict(Q=-1), dict(q=None), dict(W=None), dict(ids=None)):
        assert two(**kw) == -1 and len(L.mfar_last_error()) > 0, kw
    assert L.mfar_index_write_rows(h, F, 0, 1, P(q), 0, None) == -1            # field out of range
    assert L.mfar_index_write_rows(h, 0, D - 1, 2, P(q), 0, None) == -1        # rows past the shard
    assert L.mfar_retrieve_field(h, -1, P(q), 64, 100, 1, P(ids), P(sc), 0, None) == -1
    assert L.mfar_stage1_begin(h, P(q), 64, 100, 1, 4, P(ids), P(sc), None) == -1      # slot out of range
    assert L.mfar_set_screen(h, 3, 1.0) == -1 and L.mfar_set_stage2_dump(h, 9) == -1 and L.mfar_set_auto_off(h, 2, 0, 0) == -1
    # the pipeline
    p = ctypes.c_void_p()
    mk = lambda **kw: L.mfar_pipeline_create(ctypes.byref(p), h, P(W), 1, None, kw.get("k1", 100), 100, 1, kw.get("mb", 64), kw.get("depth", 0), kw.get("co", 0), 0)
    for kw in (dict(mb=0), dict(mb=129), dict(depth=1), dict(depth=5), dict(co=3), dict(k1=0)):
        assert mk(**kw) == -1 and not p.value, kw
    assert mk() == 0 and p.value
    t = ctypes.c_int64()
    assert L.mfar_pipeline_submit(p, P(q), 65, 0, None, ctypes.byref(t)) == -1 and L.mfar_pipeline_submit(p, P(q), 0, 0, None, ctypes.byref(t)) == -1
    assert L.mfar_pipeline_submit(p, None, 64, 0, None, ctypes.byref(t)) == -1
    assert L.mfar_pipeline_result(p, 0, P(ids), P(sc), P(nv), 0, None) == -1           # nothing submitted yet
    tickets = []
    for _ in range(9):
        assert L.mfar_pipeline_submit(p, P(q), 64, 0, None, ctypes.byref(t)) == 0
        tickets.append(t.value)
    assert L.mfar_pipeline_result(p, tickets[0], P(ids), P(sc), P(nv), 0, None) == -1  # overwritten long ago
    assert L.mfar_pipeline_result(p, tickets[-1], None, P(sc), P(nv), 0, None) == -1   # NULL output
    ids2, sc2 = np.empty_like(ids), np.empty_like(sc)
    assert L.mfar_pipeline_result(p, tickets[-1], P(ids2), P(sc2), None, 0, None) == 0 # (the held batch is launched alone)
    assert np.array_equal(ids2, ids) and np.array_equal(sc2.view(np.uint32), sc.view(np.uint32))
    assert L.mfar_pipeline_result(p, 10 ** 6, P(ids2), P(sc2), None, 0, None) == -1
    L.mfar_pipeline_destroy(p)
    ix.close()


def test_result_view_and_stream_helper():
    """The two entry points no Python wrapper uses: mfar_pipeline_result_view (device pointers into the launch's slot: what a zero-copy
    consumer reads) must show the bytes mfar_pipeline_result / _lists copy out, and mfar_stream_wait_stage1_start must order a caller's
    stream behind the begin phase of the last launch (and be a no-op before any launch)."""
    import torch
    from mfar import _native
    from mfar.data import index as idxmod
    from mfar.data.pipeline import NativePipeline
    rng = np.random.default_rng(12)
    F, D, E, Q, k = 3, 20000, 64, 40, 100
    slab, mu, W = _mk(rng, F, D, E)
    ix = idxmod.MultiFieldIndex(D, F, E, device=0)
    lib = _native.lib()
    side = torch.cuda.Stream(device=0)
    assert lib.mfar_stream_wait_stage1_start(ix._h, ctypes.c_void_p(side.cuda_stream)) == 0       # nothing launched yet
    for f in range(F):
        ix.write_rows(f, 0, slab[f])
    ix.set_screen(2)
    q = (rng.standard_normal((Q, E)) * 0.5 + mu * 2.0).astype(np.float32)
    pl = NativePipeline(ix, W, None, max_batch=64)
    t0, t1 = pl.submit(q), pl.submit(q[::-1].copy())
    assert lib.mfar_stream_wait_stage1_start(ix._h, ctypes.c_void_p(side.cuda_stream)) == 0
    side.synchronize()
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    for t, qq in ((t0, q), (t1, q[::-1])):
        r = pl.result(t)
        fid, fsc = pl.lists(t)
        ptrs = [ctypes.c_void_p() for _ in range(5)]
        assert lib.mfar_pipeline_result_view(pl._p, ctypes.c_int64(t), *[ctypes.byref(p) for p in ptrs]) == 0
        torch.cuda.synchronize()
        for p, want in zip(ptrs, (r["ids"], r["scores"], r["n_valid"], fid, fsc)):
            got = np.empty_like(want)
            assert hip.hipMemcpy(got.ctypes.data_as(ctypes.c_void_p), p, got.nbytes, 2) == 0       # hipMemcpyDeviceToHost
            assert np.array_equal(got.view(np.uint8), want.view(np.uint8))
        ref = ix.search(np.ascontiguousarray(qq), W, None)
        assert np.array_equal(r["ids"], ref["ids"]) and np.array_equal(r["scores"].view(np.uint32), ref["scores"].view(np.uint32))
    assert lib.mfar_pipeline_result_view(pl._p, ctypes.c_int64(t1 + 5), *[None] * 5) != 0                      # not a ticket
    pl.close()
    ix.close()


def test_two_indexes_from_two_host_threads():
    """Different handles may be driven from different host threads at once (include/mfar_hip.h): two threads, each with its OWN index (an
    fp32 and a bf16 one, different shapes) and its own C-ABI pipeline, created and used concurrently -- first-use initialisation included --
    every result against the synchronous search of the same index."""
    import threading
    from mfar.data import index as idxmod
    from mfar.data.pipeline import NativePipeline
    errors, barrier = [], threading.Barrier(2)

    def worker(seed, F, D, E, dtype):
        try:
            rng = np.random.default_rng(seed)
            slab, mu, W = _mk(rng, F, D, E)
            barrier.wait()                                   # both create their index (and the library's per-device state) together
            ix = idxmod.MultiFieldIndex(D, F, E, device=0, dtype=dtype)
            for f in range(F):
                ix.write_rows(f, 0, slab[f])
            ix.set_screen(2)
            qs = [(rng.standard_normal((32, E)) * 0.5 + mu * 2.0).astype(np.float32) for _ in range(40)]
            pl = NativePipeline(ix, W, None, max_batch=32)
            tickets, got = [], []
            for i, q in enumerate(qs):
                tickets.append(pl.submit(q))
                if i >= pl.lag:
                    got.append(pl.result(tickets[i - pl.lag]))
            got += [pl.result(t) for t in tickets[len(got):]]
            pl.close()
            for q, g in zip(qs, got):
                ref = ix.search(q, W, None)
                if not (np.array_equal(g["ids"], ref["ids"]) and np.array_equal(g["scores"].view(np.uint32), ref["scores"].view(np.uint32))):
                    errors.append((seed, "result differs from the synchronous search"))
                    break
            ix.close()
        except Exception as e:                               # noqa: BLE001 -- reported by the main thread
            errors.append((seed, repr(e)))

    for rep in range(4):
        threads = [threading.Thread(target=worker, args=a) for a in ((10 * rep + 1, 3, 30000, 64, "f32"), (10 * rep + 2, 5, 21000, 96, "bf16"))]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=300)
        assert not any(t.is_alive() for t in threads)
        assert not errors, (rep, errors)


def test_rows_rewritten_while_launches_are_in_flight():
    """A training loop re-encodes a field while the previous evaluation's launches are still in flight: mfar_index_write_rows between
    submits.  The write is ordered BEHIND every launch the pipeline has enqueued (the library makes the writer's stream wait
    for its own streams): those launches read the OLD rows to the end.  A batch still held for coalescing, and everything submitted
    later, is launched after the write and sees the NEW rows (screen tables, fp16 copy, gather slab rebuilt behind a device
    synchronisation).
    Made to hit the race: the launches are parked behind a long kernel on the submitting stream (their query copies queue behind it, the
    scan stream waits for the copies), the rows are rewritten from DEVICE memory on another stream that nothing blocks, and no result is
    taken in between -- without the ordering the write lands before any launch has read a row."""
    import torch
    from mfar.data import index as idxmod
    from mfar.data.pipeline import NativePipeline
    rng = np.random.default_rng(33)
    F, D, E, Q = 3, 40000, 64, 32
    slab, mu, W = _mk(rng, F, D, E)
    new_field = (rng.standard_normal((D, E)) * 0.5 + 0.3 * mu * 4.0).astype(np.float32)
    slab2 = slab.copy()
    slab2[1] = new_field
    qs = [(rng.standard_normal((Q, E)) * 0.5 + mu * 2.0).astype(np.float32) for _ in range(10)]
    dev = torch.device("cuda:0")
    for screen in (2, 0):                                    # the certified screen (rebuild behind the write) and the plain exact pass
        ix = idxmod.MultiFieldIndex(D, F, E, device=0)
        for f in range(F):
            ix.write_rows(f, 0, slab[f])
        ix.set_screen(screen)
        pl = NativePipeline(ix, W, None, max_batch=Q)
        coal = pl.coalesce
        pl.result(pl.submit(qs[0]))                          # screen built, every buffer allocated: nothing below synchronises on its own
        new_dev = torch.from_numpy(new_field).to(dev)
        q_dev = [torch.from_numpy(q).to(dev) for q in qs]
        a = torch.randn(8192, 8192, device=dev)
        torch.cuda.synchronize()
        writer = torch.cuda.Stream(device=dev)
        for _ in range(12):                                  # ~ 100 ms of matmuls ahead of the query copies
            a = a @ a * 1e-4
        tickets = [pl.submit(q) for q in q_dev[:5]]          # 5 batches: launches of (0, 1), (2, 3) enqueued, batch 4 held (coalesce 2)
        with torch.cuda.stream(writer):
            ix.write_rows(1, 0, new_dev)                     # the whole field replaced under them, nothing waited for
        tickets += [pl.submit(q) for q in q_dev[5:6]]        # (on the default stream: the library orders the launch behind the write)
        host = lambda r: {k: v.cpu().numpy() for k, v in r.items()}
        got = [host(pl.result(t)) for t in tickets]
        for q in qs[6:]:
            got.append(pl.result(pl.submit(q)))
        pl.close()
        ix.close()
        n_old = 4 if coal == 2 else 5                        # batches whose launch was enqueued before the write
        olds = [O.c_two_stage(slab, q, W, None) for q in qs[:n_old]]
        news = [O.c_two_stage(slab2, q, W, None) for q in qs]
        same = lambda g, o: np.array_equal(g["ids"], o["ids"]) and np.array_equal(g["scores"].view(np.uint32), o["scores"].view(np.uint32))
        for i in range(n_old):                               # launched before the write: the old corpus, to the last bit
            assert same(got[i], olds[i]), (screen, i, "a launch in flight read rewritten rows")
        for i in range(n_old, len(qs)):                      # held / submitted after it: the new corpus
            assert same(got[i], news[i]), (screen, i)
