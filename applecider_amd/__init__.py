"""applecider_amd — MI355X-native forward/backward hot path of AppleCiDEr.

Python host modules (same constructor signatures / sample-dict schema as
``applecider.models`` of skyportal/applecider) over a C-ABI shared library of
hand-written HIP kernels for gfx950.  See DESIGN.md and INTEGRATION.md.
"""

__version__ = "0.1.0"
