"""TEST INFRASTRUCTURE -- stands in for pycocotools==2.0.0 (environment.yml:27, not installed here) so that the
reference's src/utils.py runs unmodified; the arithmetic is oracle/annot_ref.py (published maskApi.c algorithm)."""
