"""CPU oracle for the ViT / Swin training hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is product code.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it, and there only as the checker (never as the thing measured or shipped).
The product path (``vision-transformers-pytorch_amd/``) never imports this
package and fails loudly when its HIP extension is missing.

Parity status: the upstream repository holds no tests, golden vectors or
known-answer fixtures for this path ("parity unpinned" upstream, SURVEY.md
section 4).  This oracle is therefore pinned against outputs of the reference
itself, generated in the authoring container by ``tools/gen_goldens.py``
(which imports /root/reference on CPU) and committed under ``tests/golden/``;
``tests/test_oracle_vs_golden.py`` checks every function here against them.
"""
