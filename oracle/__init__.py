"""CPU oracle for the mod_extraction hot path.

TEST INFRASTRUCTURE ONLY.  This package restates, on the CPU, the algorithms of the
reference (christhetree/mod_extraction) that the HIP kernels in ``mod_extraction_amd``
implement.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it -- never as the thing measured or shipped.  The product package
``mod_extraction_amd`` does not import anything from here and fails loudly if its HIP
extension is missing.

How each piece is pinned (details in oracle/README.md and DESIGN.md):

* ``oracle.util`` / ``oracle.modulations`` / ``oracle.fx``  -- pinned against the reference's own
  importable modules (``mod_extraction.{util,modulations,fx}``), bit-for-bit, by
  ``tests/golden/make_golden.py`` (run in the build container where ``/root/reference`` exists)
  and by the committed fixtures under ``tests/golden/``.
* ``oracle.models`` (Spectral2DCNN conv stack, LSTMEffectModel), ``oracle.losses``,
  ``oracle.lightning`` -- restated with the same core ``torch.nn`` CPU modules the reference
  uses; pinned against the reference's ``models.py`` / ``losses.py`` / ``lightning.py`` imported
  with throw-away stubs for the absent third-party packages (stubs never stand in for
  arithmetic), using the 7 shipped LSTM-64 weight files.
* mel front end (torchaudio 0.13.1), MR-STFT loss (auraloss 0.4.0), phaser (pedalboard 0.7.3 /
  JUCE) -- third-party, source absent: restated from their published algorithms,
  **PARITY UNPINNED**.
"""
