"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the NVTabular hot path.

Nothing under ``nvtabular_amd/`` may import this package.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` use it,
and only as the checker / the thing timed on the host cores.
"""
from .nvt_oracle import *  # noqa: F401,F403
