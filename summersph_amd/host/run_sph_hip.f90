! run_sph_hip.f90 -- command-line front end of the Fortran host.
!
!   run_sph_hip [ic.txt] [max_steps] [final_snapshot.txt] [sph]
!
! A fourth argument "sph" leaves out gas self-gravity, accretion and the boundary cull.
!
! With no arguments it behaves like the reference program (SUMMER_SPH.f90:934-955): reads
! 'disc_12000_2.txt' and runs to t = 1000 writing saveN.txt files.  With max_steps it runs that
! many steps, prints the dt sequence and (optionally) writes the final state to a snapshot.
program run_sph_hip
  use sph_hip_host
  implicit none
  character(len=512) :: filename, arg
  type(particle), allocatable :: bodies(:)
  type(sink), allocatable :: sinks(:)
  real(dp), allocatable :: dts(:)
  integer :: nsteps, k
  logical :: only_sph

  filename = 'disc_12000_2.txt'
  if (command_argument_count() >= 1) call get_command_argument(1, filename)
  call read_data_from_file(trim(filename), bodies, sinks)
  if (.not. allocated(bodies)) error stop 2

  if (command_argument_count() >= 2) then
    call get_command_argument(2, arg)
    read(arg, *) nsteps
    only_sph = .false.
    if (command_argument_count() >= 4) then
      call get_command_argument(4, arg)
      only_sph = trim(arg) == 'sph'
    end if
    call simulate(bodies, sinks, max_steps=nsteps, quiet=.true., dt_log=dts, sph_only=only_sph)
    do k = 0, ubound(dts, 1)
      write(*, '(A,I0,1X,ES25.17E3)') 'dt ', k, dts(k)
    end do
    if (command_argument_count() >= 3) then
      call get_command_argument(3, arg)
      call make_save(bodies, sinks, 0, trim(arg))
    end if
  else
    call simulate(bodies, sinks)
  end if
end program run_sph_hip
