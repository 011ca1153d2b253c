"""CPU ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  PARITY UNPINNED.

Two independent CPU restatements of the arithmetic behind the reference's
vectorised ``reset()``/``step()`` hot path (``breedgym/vector/vec_env.py:79-134``):

* ``oracle.c_oracle``     -- ctypes binding of ``breed_oracle.c`` (plain C, OpenMP);
* ``oracle.numpy_oracle`` -- NumPy restatement of the same *unfused* op graph the
  reference executes through chromax/JAX (gather -> uniform < r -> cumulative XOR
  -> take_along_axis -> sum(-1) @ effects).

The reference holds none of this arithmetic in its tree (it is in the un-pinned,
un-installed ``chromax``/``jax`` packages, ``pyproject.toml:24``) and its seeded
known-answer tests need chromax's bundled genome, so end-to-end parity with the
reference is **unpinned**; the JAX PRNG layer is pinned on published vectors
(``tests/test_oracle_kat.py``).  See ``breed_oracle.h`` for the full statement.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  ``breedgym_amd`` never does.
"""
