"""Drop-in ``phc`` package backed by the MI355X engine ``phc_gnn_amd``.

Import paths, constructor signatures, parameter names/shapes and forward
signatures follow bayer-science-for-a-better-life/phc-gnn so that code written
against the reference (e.g. its benchmarks/train_*.py) imports and runs
unchanged; the arithmetic of the message-passing hot path runs in hand-written
gfx950 kernels (libphc_hip.so) instead of ATen / torch-scatter.  Modules must
live on a ROCm device to be called: there is no CPU fallback.
"""
