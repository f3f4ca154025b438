"""CPU: the Step / BaseTransformer operator-API mirror behaves like src/steps/base.py."""
import pytest

from mapping_challenge_amd.steps import BaseTransformer, Dummy, Step, make_apply_transformer


class AddOne(BaseTransformer):
    def __init__(self):
        self.fitted = 0

    def fit(self, x):
        self.fitted += 1
        return self

    def transform(self, x):
        return {'x': [v + 1 for v in x]}


def test_recursive_fit_transform_adapter_and_cache(tmp_path):
    a = Step('a', AddOne(), input_data=['input'], adapter={'x': ([('input', 'x')])}, cache_dirpath=str(tmp_path), is_trainable=True)
    b = Step('b', make_apply_transformer(lambda v, w: v * w, output_name='y', apply_on=['v', 'w']), input_steps=[a],
             input_data=['input'], adapter={'v': ([('a', 'x')]), 'w': ([('input', 'w')])}, cache_dirpath=str(tmp_path))
    out = Step('output', Dummy(), input_steps=[b], adapter={'y_pred': ([('b', 'y')])}, cache_dirpath=str(tmp_path))
    data = {'input': {'x': [1, 2, 3], 'w': [2, 2, 2]}}
    assert out.fit_transform(data) == {'y_pred': [4, 6, 8]}
    assert a.transformer.fitted == 1 and a.transformer_is_cached
    assert out.transform(data) == {'y_pred': [4, 6, 8]}        # loads the cached transformer instead of fitting
    assert a.transformer.fitted == 1
    assert out.get_step('a') is a and set(out.all_steps) == {'a', 'b', 'output'}


def test_transform_without_cached_transformer_raises(tmp_path):
    a = Step('a', AddOne(), input_data=['input'], adapter={'x': ([('input', 'x')])}, cache_dirpath=str(tmp_path), is_trainable=True)
    with pytest.raises(ValueError, match='No transformer cached a'):
        a.transform({'input': {'x': [1]}})


def test_apply_transformer_input_checks():
    t = make_apply_transformer(lambda v: v)
    with pytest.raises(Exception, match='same length'):
        t.transform(a=[1, 2], b=[1])
    with pytest.raises(Exception, match='iterable'):
        t.transform(a=3)
