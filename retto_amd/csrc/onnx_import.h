// ONNX weight importer: replaces the model loading of the reference
// (/root/reference/retto-core/src/worker.rs:30-56 resolves the source,
// /root/reference/retto-core/src/worker/ort_worker.rs:120-135 hands the .onnx bytes to ONNX Runtime).
// Here the three PP-OCRv4 .onnx files are only a parameter container: a minimal protobuf reader walks
// the graph in node order, turns the parameter-carrying nodes into a stream of events (Conv,
// ConvTranspose, BatchNormalization, MatMul/Gemm + bias, LearnableAffineBlock = Mul(scalar) -> Add(scalar),
// LayerNorm = Mul(vector) -> Add(vector) or LayerNormalization) and matches that stream against the
// model manifest (the RTWB tensor list of nets.cpp, in forward order), folding BatchNorm into the
// preceding convolution.  The result is an RTWB blob, which the normal loader consumes.
//
// No real PP-OCRv4 file exists offline: the importer is validated on files written by the tests'
// own ONNX writer (tests/onnx_writer.py) in the op patterns Paddle2ONNX is known to emit.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace rt {

struct ManifestEntry {
  std::string name;        // RTWB tensor name, e.g. "det.s2.0.dw.w"
  std::vector<int> dims;   // -1 = taken from the file (rec.head.fc: the class count)
};
// MODEL_DET / MODEL_REC also accept the PP-OCRv4 SERVER graphs (ch_PP-OCRv4_server_{det,rec}_infer.onnx, BASELINE.json
// config 5): a file whose first convolution is PPHGNet's 3 -> 64 stem is matched against the MODEL_SDET / MODEL_SREC manifest.
enum ModelKind { MODEL_DET = 0, MODEL_CLS = 1, MODEL_REC = 2, MODEL_SDET = 3, MODEL_SREC = 4 };

// RTWB tensor list of a network in forward order (nets.cpp keeps it next to the layer tables).
std::vector<ManifestEntry> model_manifest(int which);

bool looks_like_rtwb(const std::vector<uint8_t>& bytes);
// Throws RtError (code 4) with the offending node / tensor on any mismatch.
std::vector<uint8_t> onnx_to_rtwb(int which, const uint8_t* data, size_t len);

}  // namespace rt
