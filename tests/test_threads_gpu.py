"""SURVEY.md 8b 'threading': the library is re-entrant across host threads -- no mutable global state besides a thread-local
error string and a thread-local plan cache; kernels are ordered by the stream each caller passes.  Two host threads drive
different convolutions on their own HIP streams at the same time and a third provokes argument errors throughout; every
result must equal the single-threaded one bit for bit and every error message must be the calling thread's own."""
import threading

import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(seed, M, Cs, Cn, dtype):
    from vlfb import hip
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, Cs, generator=g).to("cuda:0").to(dtype)
    w = (torch.randn(Cn, Cs, generator=g) * 0.05).to("cuda:0").to(dtype)
    d = hip.conv_desc(mode=hip.FPROP, dtype=hip.dtype_code(dtype), out_dtype=hip.dtype_code(dtype), N=1, Tr=1, Hr=1, Wr=M,
                      Ts=1, Hs=1, Ws=M, Cs=Cs, Cn=Cn)
    return d, x, w


def test_two_threads_two_streams_and_a_thread_of_errors():
    from vlfb import hip
    hip.lib()
    cases = [_case(1, 4096, 256, 512, torch.bfloat16), _case(2, 3000, 512, 136, torch.float16)]
    want = []
    for d, x, w in cases:
        o = torch.empty(x.shape[0], w.shape[0], device="cuda:0", dtype=x.dtype)
        hip.conv_run(d, x, w, None, o)
        torch.cuda.synchronize()
        want.append(o.clone())
    failures = []
    stop = threading.Event()

    def worker(i):
        d, x, w = cases[i]
        s = torch.cuda.Stream(device="cuda:0")
        try:
            with torch.cuda.stream(s):
                for it in range(200):
                    o = torch.empty_like(want[i])
                    hip.conv_run(d, x, w, None, o)
                    if it % 20 == 19:
                        s.synchronize()
                        if not torch.equal(o, want[i]):
                            failures.append("thread %d iteration %d: result differs" % (i, it))
                            return
            s.synchronize()
        except Exception as e:       # noqa: BLE001
            failures.append("thread %d: %r" % (i, e))

    def troublemaker():
        bad = hip.conv_desc(mode=hip.FPROP, dtype=hip.BF16, out_dtype=hip.BF16, N=1, Tr=1, Hr=1, Wr=64, Ts=1, Hs=1, Ws=64,
                            Cs=12, Cn=64)       # Cs is not a multiple of 8: an argument error with its own message
        x = torch.zeros(64 * 12, device="cuda:0", dtype=torch.bfloat16)
        while not stop.is_set():
            try:
                hip.conv_run(bad, x, x, None, x)
                failures.append("the bad descriptor was accepted")
                return
            except hip.VlfbError as e:
                if "Cs=12" not in str(e):
                    failures.append("another thread's error message leaked: %s" % e)
                    return

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    t3 = threading.Thread(target=troublemaker)
    t3.start()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    stop.set()
    t3.join()
    assert not failures, failures
