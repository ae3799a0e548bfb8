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
