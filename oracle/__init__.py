"""CPU oracle for the reconstruct-and-render hot path.  TEST INFRASTRUCTURE ONLY.

Everything under ``oracle/`` is a CPU restatement (numpy / torch-CPU / plain C)
of the reference algorithm for this path, used as the *checker* by ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``.
Nothing in the shipped package ``3danimals_amd`` imports it; the product path
runs on the HIP library only and fails loudly when that is missing.

Parity status:
* ``dmtet_ref``, ``mesh_ref``, ``skinning_ref``, ``shade_ref`` -- pinned against
  golden vectors captured from the imported reference
  (``tests/golden/make_golden.py``).
* ``raster_ref`` (rasterize / interpolate / antialias) -- **parity unpinned**:
  the arithmetic lives in nvdiffrast, which is neither vendored nor installed
  and for which the reference holds no test vectors; the oracle restates the
  published nvdiffrast operator semantics.
"""
