"""nsc_amd - MI355X-native hot path of the NSC/CMRL neural speech codec (cocosci/NSC).

The arithmetic lives in hand-written HIP kernels (nsc_amd/csrc -> libnsc_hip.so, C ABI in include/nsc_hip.h);
this package is the Python host side mirroring the reference's modules:

  nn_core_operator            conv1d / conv1d_depth / gated_bottleneck / scalar_softmax_quantization ...
  loss_terms_and_measures     mse_loss / mfcc_loss / tf_stft / quan_loss / entropy_coding_loss ...
  neural_speech_coding_module neuralSpeechCodingModule (trainer for one codec)
  cmrl                        CMRL (cascade trainer)
  engine                      explicit fwd/bwd engine used by the trainers and bench.py

There is no CPU fallback: importing works without a GPU (host logic, layouts), compute calls raise.
"""
__version__ = "0.1.0"
