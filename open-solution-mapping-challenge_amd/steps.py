"""Operator API of the reference pipeline runtime (`src/steps/base.py`): `Step`, `BaseTransformer`,
`Dummy`, plus `make_apply_transformer` (`src/utils.py:342-389`).

The reference's own `steps` package is what the HIP transformers plug into when they are dropped
into the reference tree; it cannot be imported on the GPU box (it is not vendored
there and needs sklearn.externals), so this module provides the same contract -- recursive
`fit_transform` / `transform` over a DAG of named steps, `adapter` / `unpack` input routing,
"transformer file exists => load instead of fit" caching -- for running the pipelines of
`pipelines.py` standalone.  Behaviour follows src/steps/base.py:15-286.
"""
import os
from collections.abc import Iterable
from itertools import chain

import joblib


class BaseTransformer:
    def fit(self, *args, **kwargs):
        return self

    def transform(self, *args, **kwargs):
        return NotImplementedError

    def fit_transform(self, *args, **kwargs):
        self.fit(*args, **kwargs)
        return self.transform(*args, **kwargs)

    def load(self, filepath):
        return self

    def save(self, filepath):
        joblib.dump({}, filepath)


class Dummy(BaseTransformer):
    def transform(self, **kwargs):
        return kwargs


def identity_inputs(inputs):
    return inputs[0]


class Step:
    def __init__(self, name, transformer, input_steps=None, input_data=None, adapter=None, cache_dirpath=None,
                 is_trainable=False, cache_output=False, save_output=False, load_saved_output=False,
                 save_graph=False, force_fitting=False):
        self.name, self.transformer = name, transformer
        self.input_steps = list(input_steps or [])
        self.input_data = list(input_data or [])
        self.adapter = adapter
        self.is_trainable, self.force_fitting = is_trainable, force_fitting
        self.cache_output, self.save_output, self.load_saved_output = cache_output, save_output, load_saved_output
        self.cache_dirpath = cache_dirpath
        for d in ('transformers', 'outputs', 'tmp'):
            os.makedirs(os.path.join(cache_dirpath, d), exist_ok=True)
        self.cache_filepath_step_transformer = os.path.join(cache_dirpath, 'transformers', name)
        self.save_filepath_step_output = os.path.join(cache_dirpath, 'outputs', name)
        self._cached_output = None

    # ---- graph helpers
    @property
    def all_steps(self):
        out = {}
        for s in self.input_steps:
            out.update(s.all_steps)
        out[self.name] = self
        return out

    def get_step(self, name):
        return self.all_steps[name]

    def clean_cache(self):
        for s in self.all_steps.values():
            s._cached_output = None

    @property
    def transformer_is_cached(self):
        return os.path.exists(self.cache_filepath_step_transformer)

    # ---- execution
    def _gather(self, data, method):
        inputs = {}
        for key in self.input_data:
            inputs[key] = data[key]
        for s in self.input_steps:
            inputs[s.name] = getattr(s, method)(data)
        return self.adapt(inputs) if self.adapter else self.unpack(inputs)

    def fit_transform(self, data):
        if self._cached_output is not None and not self.force_fitting:
            return self._cached_output
        if self.load_saved_output and os.path.exists(self.save_filepath_step_output) and not self.force_fitting:
            return joblib.load(self.save_filepath_step_output)
        inputs = self._gather(data, 'fit_transform')
        if self.is_trainable:
            if self.transformer_is_cached and not self.force_fitting:
                self.transformer.load(self.cache_filepath_step_transformer)
                out = self.transformer.transform(**inputs)
            else:
                out = self.transformer.fit_transform(**inputs)
                self.transformer.save(self.cache_filepath_step_transformer)
        else:
            out = self.transformer.transform(**inputs)
        return self._finish(out)

    def transform(self, data):
        if self._cached_output is not None:
            return self._cached_output
        if self.load_saved_output and os.path.exists(self.save_filepath_step_output):
            return joblib.load(self.save_filepath_step_output)
        inputs = self._gather(data, 'transform')
        if self.is_trainable:
            if not self.transformer_is_cached:
                raise ValueError('No transformer cached {}'.format(self.name))
            self.transformer.load(self.cache_filepath_step_transformer)
        return self._finish(self.transformer.transform(**inputs))

    def _finish(self, out):
        if self.cache_output:
            self._cached_output = out
        if self.save_output:
            joblib.dump(out, self.save_filepath_step_output)
        return out

    def adapt(self, step_inputs):
        adapted = {}
        for name, mapping in self.adapter.items():
            if isinstance(mapping, str):
                adapted[name] = step_inputs[mapping]
                continue
            if len(mapping) == 2:
                step_mapping, func = mapping
            elif len(mapping) == 1:
                step_mapping, func = mapping, identity_inputs
            else:
                raise ValueError('wrong mapping specified')
            adapted[name] = func([step_inputs[s][v] for s, v in step_mapping])
        return adapted

    def unpack(self, step_inputs):
        out = {}
        for d in step_inputs.values():
            out.update(d)
        return out


def make_apply_transformer(func, output_name='output', apply_on=None):
    """src/utils.py:342-389: zip the per-image inputs and call func(*args) for every image."""
    class StaticApplyTransformer(BaseTransformer):
        def transform(self, *args, **kwargs):
            self.check_input(*args, **kwargs)
            if not apply_on:
                iterator = zip(*args, *kwargs.values())
            else:
                iterator = zip(*args, *[kwargs[key] for key in apply_on])
            return {output_name: [func(*func_args) for func_args in iterator]}

        @staticmethod
        def check_input(*args, **kwargs):
            if len(args) and len(kwargs) == 0:
                raise Exception('Input must not be empty')
            length = None
            for arg in chain(args, kwargs.values()):
                if not isinstance(arg, Iterable):
                    raise Exception('All inputs must be iterable')
                try:
                    n = len(arg)
                except Exception:
                    continue
                if length is None:
                    length = n
                elif n != length:
                    raise Exception('All inputs must be the same length')
    return StaticApplyTransformer()
