"""oracle/ -- CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY.

This package is the *checker* for the MI355X path, never the product:
only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import it.  Nothing under `stable-diffusion_amd/` imports it, and the product
path raises if the HIP library is missing rather than falling back to this code.

What it restates (reference = CompVis/stable-diffusion, paths relative to the
reference root):

* `oracle.plan`        -- block layout of `UNetModel.__init__`
                          (ldm/modules/diffusionmodules/openaimodel.py:443-692)
* `oracle.weights`     -- seeded synthetic state_dict with the reference's key
                          names; zero_module tensors are
                          re-randomised (openaimodel.py:229-231,685; attention.py:244-248)
* `oracle.unet_ref`    -- `UNetModel.forward` (openaimodel.py:710-742) and every
                          module under it, as plain fp32 torch functional code
* `oracle.samplers_ref`-- PLMS / DDIM loops (ldm/models/diffusion/plms.py,
                          ddim.py) and the schedule builders (util.py:21-74)

Pinning: the reference ships no golden vectors or tests for this path
(SURVEY.md section 4), so parity is pinned the only way available: the
reference modules themselves are imported and executed in the build container
by `oracle/make_golden.py`, the oracle is asserted equal to them there, and the
resulting tensors are committed under `tests/golden/`.  `tests/test_oracle_golden.py`
re-checks the oracle against those fixtures on every run (CPU, no reference needed).
"""
