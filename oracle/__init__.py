"""CPU oracle for the Rothermel fire-spread path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``simfire_amd/`` (the product) imports this package; only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may.  Contents:

* ``rothermel_np``  - NumPy restatement of ``compute_rate_of_spread``
  (reference ``simfire/world/rothermel.py:4-136``) with every dtype made explicit.
* ``fire_sprites``  - literal sprite-list restatement of
  ``RothermelFireManager.update`` (reference ``simfire/game/managers/fire.py:616-719``)
  in pure Python/NumPy; small grids only.
* ``fire_dense.c`` / ``fire_dense`` - dense per-cell C restatement (order-free
  formulation) + libm Rothermel chain; fast enough for 1024^2 and
  used as the timed CPU baseline.

Parity is pinned: all three are checked against golden vectors generated from
the real reference (``tests/golden/make_golden.py``; fixtures in ``tests/golden``).
"""
