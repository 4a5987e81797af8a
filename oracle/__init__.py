"""oracle/ — CPU restatement of the AppleCiDEr hot path.  TEST INFRASTRUCTURE ONLY.

Nothing under applecider_amd/ may import this package.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and there only as the
checker / the CPU baseline being timed beside the GPU path — never as the product.

What it is: the reference (skyportal/applecider, pure Python on PyTorch) restated as pure
functions  f(state_dict, inputs) -> outputs  on CPU fp32 tensors, each citing the reference
file:line it follows.  state_dict keys are the reference modules' own parameter names, so the
same closed-form weights (oracle/weights.py) drive the reference, the oracle and the HIP path.

Pinning: the reference ships no numerical test for this path.  The oracle is
pinned by golden vectors generated in the build container by importing the reference itself
(tools/make_goldens.py, three in-memory stubs: applecider._version, hyrax.models.hyrax_model,
timm.create_model) and committed under tests/golden/.  One boundary stays "parity unpinned":
the inside of timm's convnext_tiny (third-party, absent, unpinned in pyproject.toml:22; archived
env pins timm 1.0.15) — restated in oracle/convnext.py from the published architecture and
cross-checked against HuggingFace transformers' independent ConvNextModel.  The archived
4-modality fusion module (_archive/AppleCider/core/model.py:8-67) is not importable as written;
its restatement (oracle/fusion.py) is validated op-by-op against torch primitives only.
"""
