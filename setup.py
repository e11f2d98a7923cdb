"""pip install -e . : builds libbigsi_hip.so for gfx950 in-tree (hipcc) and installs the bigsi_amd package."""
import os
import subprocess

from setuptools import find_packages, setup
from setuptools.command.build_py import build_py
from setuptools.command.develop import develop

HERE = os.path.dirname(os.path.abspath(__file__))


def build_hip():
    subprocess.check_call(["bash", os.path.join(HERE, "bigsi_amd", "csrc", "build.sh")])


class BuildPy(build_py):
    def run(self):
        build_hip()
        build_py.run(self)


class Develop(develop):
    def run(self):
        build_hip()
        develop.run(self)


setup(
    name="bigsi-amd",
    version=open(os.path.join(HERE, "bigsi_amd", "version.py")).read().split('"')[1],
    description="BIGSI query path on AMD MI355X (HIP)",
    packages=find_packages(include=["bigsi_amd", "bigsi_amd.*"]),
    package_data={"bigsi_amd": ["libbigsi_hip.so", "csrc/*"]},
    python_requires=">=3.8",
    install_requires=["numpy", "pyyaml"],
    cmdclass={"build_py": BuildPy, "develop": Develop},
    entry_points={"console_scripts": ["bigsi-amd=bigsi_amd.__main__:main"]},
)
