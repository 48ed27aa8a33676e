"""rip_debug_hbm_probe: the library's streaming microbenchmarks behind bench.py's `roofline.empirical` (measurement hook,
no reference counterpart)."""
import pytest

pytestmark = pytest.mark.gpu


def test_hbm_probe_reports_a_plausible_rate_for_every_access_shape(rip_lib, capsys):
    from raw_image_pipeline_amd import RawImagePipeline
    pipe = RawImagePipeline(False, "", "", "", device=0)
    rates = {k: pipe.hbm_probe(k, 256 << 20, 5) for k in pipe.PROBE_KINDS}
    with capsys.disabled():
        print("\nhbm probes, 256 MiB: " + ", ".join("%s %.0f GB/s" % kv for kv in sorted(rates.items())))
    for k, v in rates.items():
        # far above anything a mis-timed or empty launch would report, below the 8 TB/s spec of the memory
        assert 500.0 < v < 8000.0, (k, v)
    for bad in ((17, 1 << 20, 1), ("copy", 16, 1), ("copy", 1 << 20, 0)):
        with pytest.raises(ValueError):  # RIP_ERR_INVALID_ARGUMENT
            pipe.hbm_probe(*bad)
