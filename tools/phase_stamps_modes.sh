#!/bin/bash
# per-phase cycles of the forward kernel on the headline shape: automatic dispatch, then the unsharded 2-particle kernel
python tools/phase_stamps.py c1 0 && python tools/phase_stamps.py c1 2
