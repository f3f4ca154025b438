"""TEST INFRASTRUCTURE (oracle shim): pydensecrf is an unpinned git HEAD in the reference
(environment.yml:15), wraps Kraehenbuehl's densecrf C++ and is not installable here.  The API the
reference calls (src/postprocessing.py:211-223) is provided on top of oracle/crf_ref.py, an exact
(windowed, no permutohedral approximation) mean-field restatement.  PARITY UNPINNED."""
