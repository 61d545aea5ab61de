"""TEST INFRASTRUCTURE ONLY.

CPU restatements of the reference (opendilab/LightZero) hot path used as the parity checker:

* ``ctree_oracle.c``   -- plain-C restatement of ctree_efficientzero / ctree_muzero (+ cminimax)
* ``ctree.py``         -- ctypes wrapper exposing the reference's Cython module surface on top of it
* ``torch_models.py``  -- torch fp32 restatement of the EfficientZero / MuZero networks
* ``search.py``        -- restatement of EfficientZeroMCTSCtree.search / MuZeroMCTSCtree.search and of
                          the _forward_collect glue
* ``build_ref.py``     -- compiles the reference's *own* ctree sources into oracle/_ref/ (git-ignored)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
