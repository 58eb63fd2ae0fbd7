"""CPU restatement of the reference's hot path -- TEST INFRASTRUCTURE ONLY.

This package restates, in plain numpy / torch-CPU calls and C, the algorithms of tbepler/topaz
v0.3.18 that topaz_amd implements in HIP: filled ResNet8/16 and Conv127/63/31 scoring
(oracle/scoring.py), U-Net / FCNN / affine denoising with the reference's patching and
normalisation rules (oracle/denoising.py) and greedy non-maximum suppression in 2-D and 3-D
(oracle/nms.py, oracle/nms_c.c).  Every function cites the reference file:line it follows.

It is the checker, never the product: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import it.  topaz_amd never imports oracle and raises when the HIP library
is missing.

Pinning: the reference's own tests hold no numerical fixtures for this path,
so the oracle is pinned against outputs of the reference itself, generated in the build container
by oracle/make_golden.py (which imports /root/reference) and committed under tests/golden/.
tests/test_oracle_golden.py checks the oracle against every one of those vectors.
The conv / pool / interpolate arithmetic itself lives in PyTorch (torch>=1.0.0, unpinned in the
reference's requirements.txt; 2.10.0 here) and numpy's sort (numpy>=1.11; 2.2.6 here); the oracle
calls the same torch-CPU primitives, so its rounding is that of the reference's CPU path.
"""
