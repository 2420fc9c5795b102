#!/bin/sh
# Pin this repository's oracle to the REAL reference arithmetic -- one command, for whoever has a network and docker:
#
#     tools/pin/run.sh            # CPU: golden vectors (fresh-init-like AND trained-like weights) + a tf.train.Saver mini checkpoint
#     tools/pin/run.sh --gpu      # additionally, on a machine with an NVIDIA GPU: the CudnnLSTM checkpoint (GPU-trained models' format)
#
# What it does (nothing else can move parity from "unpinned" to pinned):
#   1. builds an image with tensorflow==1.13.2 / numpy==1.18.0, the versions the reference pins (/root/reference/README.md:127);
#   2. runs tools/mint_tf_golden.py in it with the repository bind-mounted: TensorFlow's OWN ops, called the way clair/model.py:299-312,
#      400-622 calls them, on recipe weights loaded BY NAME -> tests/golden/nn_tf113_64.npz, nn_tf113_trained_64.npz, tf113_mini.*
#      (and tf113_cudnn.* with --gpu);
#   3. runs the consumers on the host (python >= 3.8 with numpy and pytest, gcc for the oracle): the float32 oracle against TensorFlow's
#      vectors, the checkpoint reader against TensorFlow's own bundle bytes and variable names;
#   4. prints the files to commit.  The GPU consumer (HIP kernels against the same vectors) runs wherever an MI355X is:
#      python -m pytest tests/test_parity_gpu.py -m gpu -k tf113
# A failing consumer is the finding this kit exists for: it means this repository's reading of TensorFlow 1.13 ([TF-recall] in
# gate order i, c~, f, o; forget bias 0; slice-dense layout u*256+c; selu before softmax; variable names) is wrong somewhere --
# fix oracle/clair_oracle.c and the kernels, never the vectors.
set -e
cd "$(dirname "$0")/../.."
GPU=0; [ "$1" = "--gpu" ] && GPU=1
docker build -f tools/pin/Dockerfile.tf113 -t clair-amd-pin-tf113 tools/pin
docker run --rm -u "$(id -u):$(id -g)" -v "$PWD":/repo clair-amd-pin-tf113 --variant both --mini-checkpoint
if [ $GPU = 1 ]; then
  docker build -f tools/pin/Dockerfile.tf113-gpu -t clair-amd-pin-tf113-gpu tools/pin
  docker run --rm --gpus all -u "$(id -u):$(id -g)" -v "$PWD":/repo --entrypoint python clair-amd-pin-tf113-gpu -c "
import sys; sys.path.insert(0, 'tools'); import mint_tf_golden as m; m.cudnn_checkpoint('tests/golden/tf113_cudnn')"
fi
make -s -C oracle all
python -m pytest tests/test_oracle.py tests/test_weights.py -q -rs -k "tf113 or tensorflow or cudnn or recipe"
echo
echo "Pinned.  Commit:"
ls -1 tests/golden/nn_tf113_64.npz tests/golden/nn_tf113_trained_64.npz tests/golden/tf113_mini.index tests/golden/tf113_mini.data-00000-of-00001 tests/golden/tf113_mini.json
[ $GPU = 1 ] && ls -1 tests/golden/tf113_cudnn*
echo "then, on an MI355X:  python -m pytest tests/test_parity_gpu.py -m gpu -k tf113   and drop 'PARITY UNPINNED' from DESIGN.md section 1 / oracle/clair_oracle.c"
