"""CPU oracle for the composed-Vicuna forward/generation path.

TEST INFRASTRUCTURE ONLY.  Everything in this package is a plain torch-CPU fp32
restatement of the reference algorithm (each function cites the reference
file:line it follows).  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it, and only as the checker or
as the timed CPU baseline — never as part of the product path.  The product
package ``modelcompose_amd`` does not import it and fails loudly when the HIP
library is missing.

Parity pin: the reference ships no tests/golden vectors, so the
oracle is pinned against outputs of the reference itself, produced in the build
container by ``oracle/gen_golden.py`` (imports /root/reference through
``oracle/refshim.py``) and committed as fp32 fixtures under ``tests/golden/``.
Third-party pieces absent from /root/reference (transformers==4.31 llama/clip
math, peft==0.4.0 LoRA container) are restated from their published behaviour;
for those the pin is {shimmed reference, HF 5.15, this oracle} agreeing on the
same weights.
"""
