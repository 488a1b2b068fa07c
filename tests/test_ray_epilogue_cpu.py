"""Host-side behaviour of the fused head's lazy output dict (ponderv2_amd/ray_epilogue.py)."""


def test_render_outputs_materialise_on_demand():
    from ponderv2_amd.ray_epilogue import RenderOutputs

    calls = []

    def fill():
        calls.append(1)
        return dict(rgb=1, depth=2)

    out = RenderOutputs(dict(sdf=0), dict(comp=None), fill)
    assert out["sdf"] == 0 and not calls
    assert out.get("rgb") == 1 and len(calls) == 1
    assert "depth" in out and set(out.keys()) == {"sdf", "rgb", "depth"} and len(calls) == 1
