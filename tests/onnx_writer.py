"""An ONNX writer that does NOT share code with rm_radar_amd/onnx_import.py: the subset of onnx.proto3 an
Ultralytics export uses (ModelProto, GraphProto, NodeProto, AttributeProto, TensorProto, ValueInfoProto,
OperatorSetIdProto), declared to google.protobuf at run time and serialised by protobuf's own encoder --
packed repeated fields, field order, varint widths are protobuf's choices, as in a file written by
torch.onnx / the onnx package (which are not in this image).  Field numbers follow the public onnx.proto3."""
import numpy as np
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

_F = descriptor_pb2.FieldDescriptorProto


def _field(msg, name, number, ftype, label=_F.LABEL_OPTIONAL, type_name=None):
    f = msg.field.add()
    f.name, f.number, f.type, f.label = name, number, ftype, label
    if type_name:
        f.type_name = ".onnx_min." + type_name
    return f


def _build():
    fd = descriptor_pb2.FileDescriptorProto()
    fd.name, fd.package, fd.syntax = "onnx_min.proto", "onnx_min", "proto3"
    rep = _F.LABEL_REPEATED
    t = fd.message_type.add()
    t.name = "TensorProto"
    _field(t, "dims", 1, _F.TYPE_INT64, rep)
    _field(t, "data_type", 2, _F.TYPE_INT32)
    _field(t, "float_data", 4, _F.TYPE_FLOAT, rep)
    _field(t, "int64_data", 7, _F.TYPE_INT64, rep)
    _field(t, "name", 8, _F.TYPE_STRING)
    _field(t, "raw_data", 9, _F.TYPE_BYTES)
    _field(t, "doc_string", 12, _F.TYPE_STRING)
    a = fd.message_type.add()
    a.name = "AttributeProto"
    _field(a, "name", 1, _F.TYPE_STRING)
    _field(a, "i", 3, _F.TYPE_INT64)
    _field(a, "ints", 8, _F.TYPE_INT64, rep)
    _field(a, "type", 20, _F.TYPE_INT32)
    n = fd.message_type.add()
    n.name = "NodeProto"
    _field(n, "input", 1, _F.TYPE_STRING, rep)
    _field(n, "output", 2, _F.TYPE_STRING, rep)
    _field(n, "name", 3, _F.TYPE_STRING)
    _field(n, "op_type", 4, _F.TYPE_STRING)
    _field(n, "attribute", 5, _F.TYPE_MESSAGE, rep, "AttributeProto")
    v = fd.message_type.add()
    v.name = "ValueInfoProto"
    _field(v, "name", 1, _F.TYPE_STRING)
    g = fd.message_type.add()
    g.name = "GraphProto"
    _field(g, "node", 1, _F.TYPE_MESSAGE, rep, "NodeProto")
    _field(g, "name", 2, _F.TYPE_STRING)
    _field(g, "initializer", 5, _F.TYPE_MESSAGE, rep, "TensorProto")
    _field(g, "doc_string", 10, _F.TYPE_STRING)
    _field(g, "input", 11, _F.TYPE_MESSAGE, rep, "ValueInfoProto")
    _field(g, "output", 12, _F.TYPE_MESSAGE, rep, "ValueInfoProto")
    o = fd.message_type.add()
    o.name = "OperatorSetIdProto"
    _field(o, "domain", 1, _F.TYPE_STRING)
    _field(o, "version", 2, _F.TYPE_INT64)
    m = fd.message_type.add()
    m.name = "ModelProto"
    _field(m, "ir_version", 1, _F.TYPE_INT64)
    _field(m, "producer_name", 2, _F.TYPE_STRING)
    _field(m, "producer_version", 3, _F.TYPE_STRING)
    _field(m, "graph", 7, _F.TYPE_MESSAGE, type_name="GraphProto")
    _field(m, "opset_import", 8, _F.TYPE_MESSAGE, rep, "OperatorSetIdProto")
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    return {k: message_factory.GetMessageClass(pool.FindMessageTypeByName("onnx_min." + k))
            for k in ("ModelProto", "TensorProto")}


_CLS = None


def write_yolov8_onnx(path, tensors, forms=("raw", "float_data", "f16")):
    """tensors: name -> f32 array (conv weights with BatchNorm folded, biases), in module order.  Tensor i is
    stored in forms[i % len(forms)]: raw little-endian f32 bytes, the repeated float_data field (packed by
    protobuf), or FLOAT16 raw bytes.  One Conv node per weight and a few non-float initialisers are added,
    as a real export has."""
    global _CLS
    _CLS = _CLS or _build()
    m = _CLS["ModelProto"]()
    m.ir_version, m.producer_name, m.producer_version = 8, "pytorch", "2.x"
    op = m.opset_import.add()
    op.domain, op.version = "", 17
    g = m.graph
    g.name = "main_graph"
    g.input.add().name = "images"
    g.output.add().name = "output0"
    prev = "images"
    for i, (name, arr) in enumerate(tensors.items()):
        t = g.initializer.add()
        t.name = name
        t.dims.extend(int(d) for d in arr.shape)
        form = forms[i % len(forms)]
        if form == "f16":
            t.data_type = 10
            t.raw_data = np.ascontiguousarray(arr, "<f2").tobytes()
        elif form == "float_data":
            t.data_type = 1
            t.float_data.extend(np.ascontiguousarray(arr, np.float32).ravel().tolist())
        else:
            t.data_type = 1
            t.raw_data = np.ascontiguousarray(arr, "<f4").tobytes()
        if name.endswith(".weight"):
            node = g.node.add()
            node.op_type, node.name = "Conv", "/" + name[:-7].replace(".", "/") + "/Conv"
            node.input.extend([prev, name, name[:-6] + "bias"])
            prev = node.name + "_output_0"
            node.output.append(prev)
            ks = node.attribute.add()
            ks.name, ks.type = "kernel_shape", 7
            ks.ints.extend([int(arr.shape[2]), int(arr.shape[3])])
    for cname, vals in (("/model.22/Constant_output_0", [1, 4, 16, -1]), ("onnx::Reshape_999", [0, 144, -1])):
        c = g.initializer.add()   # int64 shape constants: not weights, must be skipped by the reader
        c.name, c.data_type = cname, 7
        c.dims.append(len(vals))
        c.int64_data.extend(vals)
    with open(path, "wb") as f:
        f.write(m.SerializeToString())
    return path
