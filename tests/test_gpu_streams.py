"""Stream / hardware-queue calibration (gnn_pretraining_amd/streams.py, libgnnmp gmp_streams_share_queue)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_a_stream_shares_its_own_queue_and_chosen_streams_run_beside_main():
    from gnn_pretraining_amd import streams as ST
    dev = torch.device("cuda:0")
    main = torch.cuda.current_stream(dev)
    assert ST.share_queue(main.cuda_stream, main.cuda_stream)
    chosen = ST.concurrent_streams(dev, 3)
    assert len({s.cuda_stream for s in chosen}) == 3 and main.cuda_stream not in {s.cuda_stream for s in chosen}
    rep = ST.last_report
    assert rep["calibrated"] and rep["own_queue"] >= 1           # at least one hardware queue beside the main stream's
    for s in chosen[:rep["own_queue"]]:                            # measured again: still beside main, and beside each other
        assert not ST.share_queue(main.cuda_stream, s.cuda_stream)
    own = chosen[:rep["own_queue"]]
    for i in range(len(own)):
        for j in range(i + 1, len(own)):
            assert not ST.share_queue(own[i].cuda_stream, own[j].cuda_stream)
    assert ST.concurrent_streams(dev, 3) == chosen               # cached: every engine of the process gets the same streams


def test_spin_kernel_waits_about_as_long_as_asked():
    from gnn_pretraining_amd import _lib as L
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    L.check(L.lib().gmp_spin_us(10, st.cuda_stream), "spin")
    torch.cuda.synchronize()
    e0.record(st)
    L.check(L.lib().gmp_spin_us(2000, st.cuda_stream), "spin")
    e1.record(st)
    torch.cuda.synchronize()
    assert 1.5 <= e0.elapsed_time(e1) <= 4.0
    assert L.lib().gmp_spin_us(-1, st.cuda_stream) != 0


def test_gate_opens_across_streams_and_waits_for_every_flag_of_its_mask():
    from gnn_pretraining_amd import _lib as L, streams as ST
    dev = torch.device("cuda:0")
    lib = L.lib()
    main = torch.cuda.current_stream(dev)
    others = ST.concurrent_streams(dev, 3)
    if ST.last_report["own_queue"] < 2:
        pytest.skip("fewer than two hardware queues beside the main stream's")
    flags = torch.zeros(64, dtype=torch.int32, device=dev)
    out = torch.zeros(4, device=dev)
    a, b = others[0], others[1]
    # b is gated on flags 3 and 5 (epoch 7); main opens 3 after a 2 ms spin, a opens 5 after a 4 ms spin
    L.check(lib.gmp_gate_wait(flags.data_ptr(), (1 << 3) | (1 << 5), 7, flags.data_ptr() + 4 * 63, b.cuda_stream), "wait")
    e_b = torch.cuda.Event(); e_b.record(b)
    L.check(lib.gmp_spin_us(2000, main.cuda_stream), "spin")
    L.check(lib.gmp_gate_open(flags.data_ptr() + 4 * 3, 7, main.cuda_stream), "open")
    e_m = torch.cuda.Event(); e_m.record(main)
    e_m.synchronize()
    assert not e_b.query()                                   # one flag of two: still closed
    L.check(lib.gmp_spin_us(2000, a.cuda_stream), "spin")
    L.check(lib.gmp_gate_open(flags.data_ptr() + 4 * 5, 8, a.cuda_stream), "open")      # a later epoch opens it too (>=)
    torch.cuda.synchronize()
    assert e_b.query()
    f = flags.cpu()
    assert int(f[3]) == 7 and int(f[5]) == 8 and int(f[63]) == 0
    assert lib.gmp_gate_wait(None, 1, 1, None, main.cuda_stream) != 0


def test_upload_kernel_moves_pinned_pieces_bit_for_bit_and_rejects_bad_arguments():
    """gmp_upload: up to four pinned host buffers -> device buffers by one kernel (the step's index arrays)."""
    import ctypes as C
    from gnn_pretraining_amd import _lib as L
    dev = torch.device("cuda:0")
    lib = L.lib()
    st = torch.cuda.current_stream(dev).cuda_stream
    g = torch.Generator().manual_seed(3)
    sizes = [4 * 12345, 2 * 777, 64, 0]                                   # elements; every piece a multiple of 16 bytes
    src = [torch.randint(-2 ** 31, 2 ** 31 - 1, (sizes[0],), dtype=torch.int32, generator=g).pin_memory(),
           torch.randint(-2 ** 62, 2 ** 62, (sizes[1],), dtype=torch.int64, generator=g).pin_memory(),
           torch.randn(sizes[2], generator=g).pin_memory(), torch.zeros(4).pin_memory()]
    dst = [torch.zeros(sizes[0] + 8, dtype=torch.int32, device=dev), torch.zeros(sizes[1] + 8, dtype=torch.int64, device=dev),
           torch.zeros(sizes[2] + 8, device=dev), torch.full((4,), 7.0, device=dev)]
    nbytes = (C.c_int64 * 4)(*[s.numel() * s.element_size() for s in src[:3]], 0)
    sp = (C.c_void_p * 4)(*[s.data_ptr() for s in src])
    dp = (C.c_void_p * 4)(*[d.data_ptr() for d in dst])
    L.check(lib.gmp_upload(4, sp, dp, nbytes, st), "gmp_upload")
    torch.cuda.synchronize()
    for s, d, n in zip(src[:3], dst[:3], sizes[:3]):
        assert torch.equal(d[:n].cpu(), s) and int(d[n:].abs().sum().item()) == 0      # nothing written past the piece
    assert torch.equal(dst[3].cpu(), torch.full((4,), 7.0))                             # an empty piece writes nothing
    bad = (C.c_int64 * 4)(24, 0, 0, 0)                                                   # not a multiple of 16
    assert lib.gmp_upload(4, sp, dp, bad, st) != 0
    assert lib.gmp_upload(5, sp, dp, nbytes, st) != 0 and lib.gmp_upload(0, sp, dp, nbytes, st) != 0
