"""
Recursive update plumbing, same contract as tfrt/update.py:3-78: ``update()`` runs the
``update_handles`` (children first), then ``_update()``, then ``post_update_handles``;
``frozen`` disables all of it; ``forced_update()`` ignores ``frozen`` (and, like the
reference, does not run the post handles).
"""
from abc import ABC, abstractmethod


class RecursivelyUpdatable(ABC):
    def __init__(self, update_handles=None, recursively_update=True, frozen=False, **kwargs):
        self.recursively_update = recursively_update
        self.frozen = False
        self.update_handles = (
            self._generate_update_handles() if update_handles is None else update_handles
        )
        self.post_update_handles = []
        self.update()  # the reference updates once at construction (update.py:50)
        self.frozen = frozen

    def update(self):
        if self.frozen:
            return
        if self.recursively_update and bool(self.update_handles):
            for handle in self.update_handles:
                handle()
        self._update()
        for handle in self.post_update_handles:
            handle()

    def forced_update(self):
        if self.recursively_update:
            for handle in self.update_handles:
                handle()
        self._update()

    @abstractmethod
    def _update(self):
        raise NotImplementedError

    @abstractmethod
    def _generate_update_handles(self):
        raise NotImplementedError
