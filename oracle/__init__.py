"""CPU oracle for the DP-CGAN training-step hot path.

TEST INFRASTRUCTURE ONLY.  This package restates, on the CPU (numpy fp64/int64 for the sampler and
encode, torch-CPU fp32 autograd for the networks), what the reference
sunchang0124/dp_cgans computes on the path SURVEY.md section 8 lists.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it; the product
package ``dp_cgans_amd`` never does and fails loudly when its HIP library is missing.

Parity is PINNED: every function here is checked in ``tests/test_oracle_golden.py`` against
fixtures under ``tests/golden/`` that were captured from the imported reference itself by
``tools/capture_goldens.py`` (two full training steps, private and non-private, the sampler, the
encode, all with recorded random draws).  The reference's own tests hold no numeric results
(SURVEY.md F10), so those captures are the pin.
"""
from . import nets, sampler, encode  # noqa: F401
