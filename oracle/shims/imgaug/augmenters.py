"""TEST INFRASTRUCTURE (oracle shim) -- see imgaug/__init__.py.  Augmenter protocol: reseed(), to_deterministic(),
augment_image(image); subclasses of the reference (PadFixed, RandomCropFixedSize) implement _augment_images."""
import numpy as np


class Augmenter:
    def __init__(self, name=None, deterministic=False, random_state=None):
        self.name, self.deterministic = name, deterministic
        self.random_state = random_state if random_state is not None else np.random.RandomState(1234)

    def reseed(self, *a, **k):
        return self

    def to_deterministic(self):
        return self

    def augment_image(self, image):
        return self.augment_images([image])[0]

    def augment_images(self, images):
        if type(self)._augment_images is not Augmenter._augment_images:
            return self._augment_images(images, self.random_state, [], None)
        return list(images)

    def _augment_images(self, images, random_state, parents, hooks):
        return list(images)


class Sequential(Augmenter):
    def __init__(self, children=None, random_order=False, **kw):
        super().__init__(**kw)
        self.children = list(children or [])

    def augment_images(self, images):
        for c in self.children:
            images = c.augment_images(images) if isinstance(c, Augmenter) else images
        return list(images)


class _Identity(Augmenter):
    def __init__(self, *a, **k):
        super().__init__()


class SomeOf(_Identity):
    pass


class OneOf(_Identity):
    pass


class Sometimes(_Identity):
    pass


class Fliplr(_Identity):
    pass


class Flipud(_Identity):
    pass


class Affine(_Identity):
    pass


class ChangeColorspace(_Identity):
    pass


class WithChannels(_Identity):
    pass


class Add(_Identity):
    pass


class Noop(_Identity):
    pass
