"""CPU oracle for the RNN-T hot path — TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package; the product (``caiman_asr_amd``) never does.

``oracle.native``  : ctypes front-end of ``rnnt_oracle.c`` (C restatement of the
                     reference's CUDA operators, double precision).
``oracle.brute``   : path-enumeration definition of the transducer loss (pure
                     Python, tiny lattices only) used to pin ``native``.
``oracle.model``   : torch-CPU restatement of the RNNT network / decoders, pinned
                     by golden vectors generated from the reference's own Python
                     (``oracle/gen_golden.py`` -> ``tests/golden``).
"""
