"""not gpu: error behaviour and symbolic (build-pass) shape arithmetic of the drop-in op API — same exceptions as the reference where it
raises (layers.simple_concat2d's ValueError), ValueError for the shape errors TF would report at graph construction."""
import pytest
import torch

from conftest import pkg


@pytest.fixture()
def store():
    V = pkg("variables")
    st = V.VariableStore("cpu", seed=0)
    with st.as_default():
        st.begin_trace()
        yield st


def meta(*shape):
    return torch.empty(shape, device="meta")


def test_shape_arithmetic_on_the_build_pass(store):
    L, O = pkg("layers"), pkg("ops")
    w = L.weight_variable([3, 3, 16, 32])
    assert tuple(L.conv2d(meta(2, 64, 64, 16), w, 1.0).shape) == (2, 64, 64, 32)
    assert tuple(L.conv2d(meta(2, 64, 64, 16), w, 1.0, strides=[1, 2, 2, 1]).shape) == (2, 32, 32, 32)
    assert tuple(L.conv2d(meta(2, 64, 64, 16), w, 1.0, padding="SYMMETRIC").shape) == (2, 64, 64, 32)
    w5 = L.weight_variable([5, 5, 16, 8])
    assert tuple(L.conv_bn_relu2d(meta(2, 16, 16, 16), w5, 1.0, strides=[1, 4, 4, 1], scope="a", leak=True).shape) == (2, 4, 4, 8)
    assert tuple(L.conv_bn_relu2d(meta(2, 4, 4, 16), L.weight_variable([3, 3, 16, 8]), 1.0, strides=[1, 2, 2, 1], padding="SYMMETRIC",
                                  scope="b", leak=True).shape) == (2, 2, 2, 8)
    assert tuple(L.max_pool2d(meta(2, 64, 64, 16), 2).shape) == (2, 32, 32, 16)
    assert tuple(O.PS(meta(2, 32, 32, 2560), r=8, n_channel=40, batch_size=2).shape) == (2, 256, 256, 40)
    w1, w2 = L.weight_variable([3, 3, 16, 32]), L.weight_variable([3, 3, 32, 32])
    assert tuple(L.residual_block(meta(2, 8, 8, 16), w1, w2, 1.0, inc_dim=True, leak=True).shape) == (2, 8, 8, 32)
    # variables created on the way carry TF names
    assert "a/gamma" in store.vars and "b/moving_variance" in store.vars and "BatchNorm_1/beta" in store.vars
    assert [n for n in store.vars if n.startswith("Variable")][:3] == ["Variable", "Variable_1", "Variable_2"]


def test_errors(store):
    L, O = pkg("layers"), pkg("ops")
    w = L.weight_variable([3, 3, 16, 32])
    with pytest.raises(ValueError):
        L.conv2d(meta(2, 8, 8, 8), w, 1.0)                       # channel mismatch
    with pytest.raises(ValueError):
        L.conv2d(meta(2, 8, 8, 16), w, 1.0, strides=[1, 2, 1, 1])   # non-square stride
    with pytest.raises(ValueError):
        L.conv2d(meta(2, 8, 8, 16), w, 1.0, padding="REFLECT")
    with pytest.raises(ValueError):
        O.PS(meta(2, 4, 4, 100), r=8, n_channel=2)               # 100 != 2*64
    with pytest.raises(ValueError):
        L.simple_concat2d(torch.zeros(1, 4, 4, 2), torch.zeros(1, 4, 5, 2))
    with pytest.raises(ValueError):
        L.max_pool2d(meta(1, 4, 4, 2), 3)
    w1, w2 = L.weight_variable([3, 3, 16, 16]), L.weight_variable([3, 3, 16, 24])
    with pytest.raises(ValueError):
        L.residual_block(meta(2, 8, 8, 16), w1, w2, 1.0, inc_dim=True)   # 16 + 2*8 != 24
    with pytest.raises(NotImplementedError):
        L.conv_relu2d(meta(1, 4, 4, 16), w, 1.0)                 # dead helper of the reference (layers.py:77-82)
    for name in ("avg_pool2d", "crop_and_concat", "pixel_wise_softmax", "cross_entropy"):      # the other dead helpers: exported, loud
        with pytest.raises(NotImplementedError):
            getattr(L, name)(meta(1, 4, 4, 2), 2)
    assert tuple(L.bias_variable([8]).shape) == (8,) and tuple(L.weight_variable_deconv([2, 2, 4, 4]).shape) == (2, 2, 4, 4)
    V = pkg("variables")
    store.finalize()
    with pytest.raises(RuntimeError):
        L.weight_variable([3, 3, 4, 4])                          # variable creation after the graph is finalised


def test_no_store_is_loud():
    L = pkg("layers")
    with pytest.raises(RuntimeError):
        L.weight_variable([3, 3, 4, 4])
