"""CPU oracle for the ptychography hot path -- TEST INFRASTRUCTURE ONLY.

Nothing in the product package (``tike_amd``) may import this package.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg use it, and there only as the checker / baseline.
"""
