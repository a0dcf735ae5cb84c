"""CPU oracle for the FruitNeRF ray-marching hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``fruitnerf_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and only as the checker.

PARITY: PARTLY PINNED.  ``oracle/fruit_oracle.py`` (everything FruitNeRF-specific)
is pinned against the reference's OWN classes and functions, executed in this
container over ``oracle/ns_torch.py`` with the uninstallable third-party imports
stubbed (``tests/golden/make_reference_*golden.py`` -> ``tests/test_reference_pins.py``);
``oracle/cloud.py``'s DBSCAN is scikit-learn itself.  ``oracle/ns_torch.py`` — the
nerfstudio 0.3.2 components — and the Open3D restatements in ``oracle/cloud.py``
remain PARITY UNPINNED: the reference (meyerls/FruitNeRF) ships no tests, golden
vectors or checkpoints, and the arithmetic of its hot path lives in
``nerfstudio==0.3.2`` (pinned at /root/reference/pyproject.toml:10), which is
neither vendored in the reference tree nor installable in the build container.
``oracle/ns_torch.py`` therefore restates nerfstudio 0.3.2's *torch-fallback*
code path (what the reference executes on CPU when tinycudann is absent) from
its published algorithm; ``oracle/fruit_oracle.py`` restates the in-tree
FruitNeRF code on top of it, citing reference file:line for every function.
Pins we create ourselves: closed-form checks in ``tests/test_oracle_*.py`` and
seeded golden vectors under ``tests/golden/`` (generator committed).
"""
