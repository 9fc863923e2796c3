"""GPU test of the arithmetic building block behind the integrate kernel's bit-exactness:
v_rcp_f32 + one Newton step is the correctly rounded reciprocal for every fp32 mantissa on this GPU,
and the 3-instruction Markstein quotient built on it equals IEEE '/'."""
import pytest

pytestmark = pytest.mark.gpu


def test_refined_reciprocal_and_markstein_quotient_are_ieee(gpu_ctx):
    n_rcp, bad_rcp, n_div, bad_div = gpu_ctx.selftest_division()
    assert n_rcp == 8 << 23 and n_div > 10 ** 9
    assert bad_rcp == 0, "%d of %d refined reciprocals differ from 1.0f/d" % (bad_rcp, n_rcp)
    assert bad_div == 0, "%d of %d quotients differ from n/d" % (bad_div, n_div)
