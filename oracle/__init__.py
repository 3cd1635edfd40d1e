"""CPU oracle for the event_flow hot path  --  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (numpy for the integer / index work, plain
PyTorch-CPU fp32 for the floating-point network and loss) of the arithmetic of
tudelft/event_flow's hot path:

  dataloader/encodings.py  ->  oracle/encodings.py
  utils/iwe.py             ->  oracle/iwe.py
  loss/flow.py             ->  oracle/loss.py
  models/spiking_util.py, spiking_submodules.py, submodules.py (ConvLayer,
  ConvLayer_, ConvGRU), model.py (FireNet family), unet.py
  (SpikingMultiResUNetRecurrent)  ->  oracle/snn.py
  train_flow.py:141-171 (loss / backward / clip / Adam)  ->  oracle/train.py

Every function cites the reference file:line it follows.  The reference ships
no tests and no golden vectors, so the oracle is PINNED
against the reference itself: tools/gen_golden.py imports /root/reference in
the build container, runs both, and commits the reference's outputs as
tests/golden/*.npz; tests/test_oracle_golden.py re-checks the oracle against
those fixtures everywhere (the reference itself never travels).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this package.  Nothing under event_flow_amd/ imports it: the product
path is the HIP library and raises if that library is missing.
"""
