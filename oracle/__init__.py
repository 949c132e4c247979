"""CPU oracle for the Kronecker-structured collapsed-ELBO hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and there only as the checker / the timed CPU baseline.  The
product package (``variational_gridded_gaussian_processes_amd``) never imports
this package and fails loudly when its HIP library is missing.

PARITY UNPINNED: the reference (maxnorman569/Variational-Gridded-Gaussian-
Processes) ships no tests, no golden vectors and no stored outputs, and its model
classes import ``gpytorch`` / ``linear_operator`` (un-vendored, un-pinned, not
installed here, no network).  The oracle is therefore this package's own float64
restatement of the reference formulas (file:line cited per function), with the six
gpytorch/linear_operator calls on the path replaced by their documented
definitions (see ``dense.py`` header).  What *is* pinned against reference code
that imports here (``src/basis/bspline.py``, ``src/utils/datagenerators.py``,
``src/utils/integrators.py``) is the mesh bookkeeping, the point ordering and the
quad known-answer check -- see ``tests/golden/make_golden.py``.

Modules
  dense.py  literal dense O(N^3) restatement (torch float64, autograd gradients)
  kron.py   structured O(N m) CPU twin of the GPU algorithm (numpy float64,
            analytic gradients) -- also the timed ``cpu_baseline`` ("port")
"""
