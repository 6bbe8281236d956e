"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference hot path.

Nothing under ``dgps_with_iwvi_amd/`` may import this package.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` use it,
and only as the checker / the reported CPU baseline.

Parity status: **parity unpinned**.  The reference (TensorFlow 1.x + GPflow 1.x)
cannot be imported in the build container or on the GPU box and holds no golden
vectors, so the restatement is pinned by closed-form
identities instead (tests/test_oracle_*.py) and by an independent unwhitened
SVGP derivation (oracle/svgp_closed_form.py).
"""
