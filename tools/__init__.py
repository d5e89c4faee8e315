"""Import-level compatibility with the reference's ``tools`` package: ``from tools.final_common import ...`` and
``from tools.final_util import ...`` resolve to the HIP-backed implementations in ``interpret_quality_amd``.
(The other files in this directory are development helpers of this repository.)"""
