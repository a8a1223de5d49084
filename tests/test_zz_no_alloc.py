"""slim_quant_search_noalloc (the entry point that allocates nothing: its search workspace is a static device array of
the library) against slim_quant_search, through the C ABI, bit for bit, on two calls in a row with the second on a
smaller histogram (fewer bins: it must not see what the first call left in the shared array)."""
import pytest
import torch

from conftest import synth_weight

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _search(lib, name, hist, bins, bits, ws=None, check_no_alloc=False):
    out = torch.full((2,), float("nan"), dtype=torch.float32, device=DEV)
    idx = torch.full((1,), -1, dtype=torch.int32, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    if check_no_alloc:
        # the outputs exist already: from here to the end of the call neither torch's allocator nor the device's free memory
        # may move (a hipMalloc inside the library would lower the latter)
        torch.cuda.synchronize()
        free0, used0 = torch.cuda.mem_get_info()[0], torch.cuda.memory_allocated()
        lib.call(name, hist.data_ptr(), bins, bits, out.data_ptr(), idx.data_ptr(), out[1:].data_ptr(), st)
        torch.cuda.synchronize()
        assert torch.cuda.memory_allocated() == used0
        assert torch.cuda.mem_get_info()[0] == free0, (free0, torch.cuda.mem_get_info()[0])
        return out.cpu(), int(idx.item())
    if ws is None:
        lib.call(name, hist.data_ptr(), bins, bits, out.data_ptr(), idx.data_ptr(), out[1:].data_ptr(), st)
    else:
        lib.call(name, hist.data_ptr(), bins, bits, out.data_ptr(), idx.data_ptr(), out[1:].data_ptr(), ws.data_ptr(),
                 ws.numel(), st)
    torch.cuda.synchronize()
    return out.cpu(), int(idx.item())


def test_noalloc_search_equals_workspace_search():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from slim_amd import _lib, ops
    # large first (2359 bins), then small (512 bins), then the largest supported (20000 bins) and small again
    cases = [(1536, 1536, 11, 4), (256, 512, 12, 4), (4608, 4608, 13, 4), (300, 1000, 14, 8)]
    for n, k, seed, bits in cases:
        W = synth_weight(n, k, seed, heavy=(seed % 2 == 0)).to(DEV)
        bins = max(512, min(W.numel() // 1000, 20000))
        hist = ops.hist_abs(W)
        ws = torch.empty(_lib.query("slim_quant_search_workspace_bytes", bins), dtype=torch.uint8, device=DEV)
        ref, ref_idx = _search(_lib, "slim_quant_search", hist, bins, bits, ws)
        # (the first call on the device resolves the static array's address and loads the kernels; later calls are checked)
        got, got_idx = _search(_lib, "slim_quant_search_noalloc", hist, bins, bits, check_no_alloc=(n, k) != cases[0][:2])
        assert got_idx == ref_idx and torch.equal(got.view(torch.int32), ref.view(torch.int32)), (n, k, got, ref)
        assert got_idx > 0 and float(got[0]) > 0
    # beyond the static scratch: refused with a clear message, nothing launched
    hist = ops.hist_abs(synth_weight(64, 64, 1).to(DEV))
    with pytest.raises(ValueError, match="exceeds the 20000 bins"):
        _search(_lib, "slim_quant_search_noalloc", hist, 20001, 4)
    with pytest.raises(ValueError, match="at least 10"):
        _search(_lib, "slim_quant_search_noalloc", hist, 5, 4)
